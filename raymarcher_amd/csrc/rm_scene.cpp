// rm_scene.cpp — scenefile loader kept from the reference: JSON scenefile → flattened object / light
// tables in the layout the render ABI consumes.  Own JSON reader (no Qt on the GPU box).
//
// Reference counterparts: ScenefileReader::readJSON and helpers (src/utils/scenefilereader.cpp:64-1153,
// schema in SURVEY Appendix B), SceneParser::parse / parseHelper / getLocTransMat
// (src/utils/sceneparser.cpp:15-133), RayMarchScene::initScene (src/raymarch/raymarchscene.cpp:104-134)
// and the per-object uniform derivation of configureShapesUniforms (src/realtimerender.cpp:732-811).
// Error behaviour: the reference prints a message and returns false (ignored upstream,
// raymarchscene.cpp:111); here every schema violation is RM_ERR_PARSE with the message in rm_last_error().
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/raymarcher_amd.h"
#include "rm_internal.h"
#include "rm_mat4.h"

namespace rm {
namespace {

// ------------------------------------------------------------------------------------------- JSON
struct JValue;
using JObject = std::vector<std::pair<std::string, JValue>>;
struct JValue {
  enum Type { Null, Bool, Number, String, Array, Object } type = Null;
  bool b = false;
  double num = 0.0;
  std::string str;
  std::vector<JValue> arr;
  JObject obj;
  bool isNumber() const { return type == Number; }
  const JValue *find(const std::string &k) const {
    const JValue *hit = nullptr;
    for (auto &kv : obj)
      if (kv.first == k) hit = &kv.second;  // last duplicate wins, as in QJsonObject
    return hit;
  }
  bool has(const std::string &k) const { return find(k) != nullptr; }
};

struct JParser {
  const char *p, *end;
  std::string err;
  explicit JParser(const std::string &s) : p(s.data()), end(s.data() + s.size()) {}
  void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++; }
  bool fail(const std::string &m) { if (err.empty()) err = m; return false; }
  bool lit(const char *s) {
    size_t n = std::strlen(s);
    if ((size_t)(end - p) < n || std::memcmp(p, s, n) != 0) return false;
    p += n;
    return true;
  }
  bool string(std::string &out) {
    if (p >= end || *p != '"') return fail("expected string");
    p++;
    while (p < end && *p != '"') {
      if (*p == '\\') {
        p++;
        if (p >= end) return fail("bad escape");
        switch (*p) {
          case '"': out += '"'; break;
          case '\\': out += '\\'; break;
          case '/': out += '/'; break;
          case 'b': out += '\b'; break;
          case 'f': out += '\f'; break;
          case 'n': out += '\n'; break;
          case 'r': out += '\r'; break;
          case 't': out += '\t'; break;
          case 'u': {
            if (end - p < 5) return fail("bad \\u escape");
            unsigned cp = (unsigned)std::strtoul(std::string(p + 1, p + 5).c_str(), nullptr, 16);
            p += 4;
            if (cp < 0x80) out += (char)cp;
            else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
            else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
            break;
          }
          default: return fail("bad escape");
        }
        p++;
      } else {
        out += *p++;
      }
    }
    if (p >= end) return fail("unterminated string");
    p++;
    return true;
  }
  bool value(JValue &v, int depth = 0) {
    if (depth > 256) return fail("nesting too deep");
    ws();
    if (p >= end) return fail("unexpected end of input");
    if (*p == '{') {
      v.type = JValue::Object;
      p++;
      ws();
      if (p < end && *p == '}') { p++; return true; }
      for (;;) {
        ws();
        std::string k;
        if (!string(k)) return false;
        ws();
        if (p >= end || *p != ':') return fail("expected ':'");
        p++;
        JValue child;
        if (!value(child, depth + 1)) return false;
        v.obj.emplace_back(std::move(k), std::move(child));
        ws();
        if (p < end && *p == ',') { p++; continue; }
        if (p < end && *p == '}') { p++; return true; }
        return fail("expected ',' or '}'");
      }
    }
    if (*p == '[') {
      v.type = JValue::Array;
      p++;
      ws();
      if (p < end && *p == ']') { p++; return true; }
      for (;;) {
        JValue child;
        if (!value(child, depth + 1)) return false;
        v.arr.push_back(std::move(child));
        ws();
        if (p < end && *p == ',') { p++; continue; }
        if (p < end && *p == ']') { p++; return true; }
        return fail("expected ',' or ']'");
      }
    }
    if (*p == '"') { v.type = JValue::String; return string(v.str); }
    if (lit("true")) { v.type = JValue::Bool; v.b = true; return true; }
    if (lit("false")) { v.type = JValue::Bool; v.b = false; return true; }
    if (lit("null")) { v.type = JValue::Null; return true; }
    if (*p == '-' || (*p >= '0' && *p <= '9')) {
      char *e = nullptr;
      v.num = std::strtod(p, &e);
      if (e == p || e > end) return fail("bad number");
      p = e;
      v.type = JValue::Number;
      return true;
    }
    return fail(std::string("unexpected character '") + *p + "'");
  }
  bool document(JValue &v) {
    if (!value(v)) return false;
    ws();
    if (p != end) return fail("trailing characters after document");
    return true;
  }
};

// ------------------------------------------------------------------------------------------- math
using M4 = Mat4;  // rm_mat4.h
inline M4 identity() { return mat_identity(); }
inline M4 mul(const M4 &A, const M4 &B) { return mat_mul(A, B); }
inline void mulVec(const M4 &A, const float v[4], float out[4]) { mat_mul_vec(A, v, out); }
inline M4 inverse(const M4 &M) { return mat_inverse(M); }
// Rodrigues rotation about normalize(axis) by `angle` radians.
M4 rotation(float angle, const float axis[3]) {
  const float c = std::cos(angle), s = std::sin(angle);
  const float l = std::sqrt(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2]);
  const float il = 1.0f / l;
  const float a[3] = {axis[0] * il, axis[1] * il, axis[2] * il};
  const float t[3] = {(1.0f - c) * a[0], (1.0f - c) * a[1], (1.0f - c) * a[2]};
  M4 R = identity();
  R.m[0] = c + t[0] * a[0];        R.m[1] = t[0] * a[1] + s * a[2]; R.m[2] = t[0] * a[2] - s * a[1];
  R.m[4] = t[1] * a[0] - s * a[2]; R.m[5] = c + t[1] * a[1];        R.m[6] = t[1] * a[2] + s * a[0];
  R.m[8] = t[2] * a[0] + s * a[1]; R.m[9] = t[2] * a[1] - s * a[0]; R.m[10] = c + t[2] * a[2];
  return R;
}

// ------------------------------------------------------------------------------------------- scene graph
struct Light {
  int type = 0;
  float color[4] = {0, 0, 0, 0};
  float func[3] = {1, 0, 0};
  float dir[4] = {0, 0, 0, 0};
  float penumbra = 0, angle = 0, width = 0, height = 0, intensity = 0;
};
struct Primitive {
  int type = RM_CUBE;
  float amb[3] = {0, 0, 0}, dif[3] = {1, 1, 1}, spec[3] = {0, 0, 0}, refl[3] = {0, 0, 0}, transp[3] = {0, 0, 0};
  float shininess = 0, ior = 0, blend = 0;
  bool textured = false;
  std::string texFile;
  float repeatU = 0, repeatV = 0;
};
struct Node {
  bool hasT = false, hasR = false, hasS = false, hasM = false;
  float translate[3] = {0, 0, 0}, rotAxis[3] = {0, 0, 0}, rotAngle = 0, scale[3] = {1, 1, 1};
  M4 matrix = identity();
  std::vector<Light> lights;
  std::vector<Primitive> prims;
  std::vector<Node *> children;  // may alias template nodes
};

const double kPi = 3.14159265358979323846;

}  // namespace
}  // namespace rm

using namespace rm;

struct RmScene {
  std::vector<RmObject> objects;
  std::vector<RmLight> lights;
  std::vector<std::string> textures;  // per object ("" if none)
  float ka = 0, kd = 0, ks = 0, kt = 0;
  RmCameraData cam{};
};

namespace rm {
namespace {

struct Reader {
  std::string baseDir;  // grand-parent directory of the scenefile (scenefilereader.cpp:940-941)
  std::vector<std::unique_ptr<Node>> nodes;
  std::map<std::string, Node *> templates;
  std::string err;

  bool fail(const std::string &m) { if (err.empty()) err = m; return false; }

  bool onlyKeys(const JValue &o, const char *what, std::initializer_list<const char *> allowed) {
    for (auto &kv : o.obj) {
      bool ok = false;
      for (auto a : allowed) ok = ok || kv.first == a;
      if (!ok) return fail("unknown field \"" + kv.first + "\" on " + what);
    }
    return true;
  }
  bool vec(const JValue &o, const char *key, int n, float *out, const char *what) {
    const JValue *v = o.find(key);
    if (!v || v->type != JValue::Array || (int)v->arr.size() != n)
      return fail(std::string(what) + " " + key + " must be an array of " + std::to_string(n) + " numbers");
    for (int i = 0; i < n; i++) {
      if (!v->arr[i].isNumber()) return fail(std::string(what) + " " + key + " must contain numbers");
      out[i] = (float)v->arr[i].num;
    }
    return true;
  }
  bool num(const JValue &o, const char *key, double *out, const char *what) {
    const JValue *v = o.find(key);
    if (!v || !v->isNumber()) return fail(std::string(what) + " " + key + " must be a number");
    *out = v->num;
    return true;
  }

  // scenefilereader.cpp:148-201
  bool globalData(const JValue &g, RmScene &sc) {
    if (!onlyKeys(g, "globalData object", {"ambientCoeff", "diffuseCoeff", "specularCoeff", "transparentCoeff"})) return false;
    double v;
    if (!g.has("ambientCoeff") || !g.has("diffuseCoeff") || !g.has("specularCoeff"))
      return fail("missing required field on globalData object");
    if (!num(g, "ambientCoeff", &v, "globalData")) return false;
    sc.ka = (float)v;
    if (!num(g, "diffuseCoeff", &v, "globalData")) return false;
    sc.kd = (float)v;
    if (!num(g, "specularCoeff", &v, "globalData")) return false;
    sc.ks = (float)v;
    if (g.has("transparentCoeff")) {
      if (!num(g, "transparentCoeff", &v, "globalData")) return false;
      sc.kt = (float)v;
    }
    return true;
  }

  // scenefilereader.cpp:454-602
  bool cameraData(const JValue &c, RmScene &sc) {
    if (!onlyKeys(c, "cameraData object", {"position", "up", "heightAngle", "aperture", "focalLength", "look", "focus"}))
      return false;
    if (!c.has("position") || !c.has("up") || !c.has("heightAngle")) return fail("missing required field on cameraData object");
    if (c.has("look") && c.has("focus")) return fail("cameraData cannot contain both \"look\" and \"focus\"");
    if (!c.has("look") && !c.has("focus")) return fail("cameraData needs \"look\" or \"focus\"");
    RmCameraData &cd = sc.cam;
    if (!vec(c, "position", 3, cd.pos, "cameraData")) return false;
    cd.pos[3] = 1.0f;
    if (!vec(c, "up", 3, cd.up, "cameraData")) return false;
    cd.up[3] = 0.0f;
    double ha;
    if (!num(c, "heightAngle", &ha, "cameraData")) return false;
    cd.heightAngle = (float)(ha * kPi / 180.0);
    double dummy;
    if (c.has("aperture") && !num(c, "aperture", &dummy, "cameraData")) return false;
    if (c.has("focalLength") && !num(c, "focalLength", &dummy, "cameraData")) return false;
    if (c.has("look")) {
      if (!vec(c, "look", 3, cd.look, "cameraData")) return false;
      cd.look[3] = 0.0f;
    } else {
      if (!vec(c, "focus", 3, cd.look, "cameraData")) return false;
      cd.look[3] = 1.0f;
      for (int i = 0; i < 4; i++) cd.look[i] = cd.look[i] - cd.pos[i];  // look = focus − position
    }
    return true;
  }

  // scenefilereader.cpp:206-449
  bool light(const JValue &l, Node *node) {
    if (!onlyKeys(l, "light object", {"type", "color", "name", "attenuationCoeff", "direction", "penumbra", "angle",
                                      "width", "height", "intensity"}))
      return false;
    if (!l.has("type") || !l.has("color")) return fail("missing required field on light object");
    Light li;
    if (!vec(l, "color", 3, li.color, "light")) return false;
    const JValue *t = l.find("type");
    if (t->type != JValue::String) return fail("light type must be a string");
    double v;
    if (t->str == "directional") {
      li.type = RM_LIGHT_DIRECTIONAL;
      if (!vec(l, "direction", 3, li.dir, "directional light")) return false;
    } else if (t->str == "point") {
      li.type = RM_LIGHT_POINT;
      if (!vec(l, "attenuationCoeff", 3, li.func, "point light")) return false;
    } else if (t->str == "spot") {
      li.type = RM_LIGHT_SPOT;
      if (!vec(l, "direction", 3, li.dir, "spot light")) return false;
      if (!vec(l, "attenuationCoeff", 3, li.func, "spot light")) return false;
      if (!num(l, "penumbra", &v, "spot light")) return false;
      li.penumbra = (float)(v * kPi / 180.0);
      if (!num(l, "angle", &v, "spot light")) return false;
      li.angle = (float)(v * kPi / 180.0);
    } else if (t->str == "area") {
      li.type = RM_LIGHT_AREA;
      if (!num(l, "width", &v, "area light")) return false;
      li.width = (float)v;
      if (!num(l, "height", &v, "area light")) return false;
      li.height = (float)v;
      if (!num(l, "intensity", &v, "area light")) return false;
      li.intensity = (float)v;
      if (!vec(l, "attenuationCoeff", 3, li.func, "area light")) return false;
    } else {
      return fail("unknown light type \"" + t->str + "\"");
    }
    node->lights.push_back(li);
    return true;
  }

  // scenefilereader.cpp:889-1153
  bool primitive(const JValue &p, Node *node) {
    if (!onlyKeys(p, "primitive object", {"type", "meshFile", "ambient", "diffuse", "specular", "reflective", "transparent",
                                          "shininess", "ior", "blend", "textureFile", "textureU", "textureV",
                                          "bumpMapFile", "bumpMapU", "bumpMapV"}))
      return false;
    const JValue *t = p.find("type");
    if (!t) return fail("missing required field \"type\" on primitive object");
    if (t->type != JValue::String) return fail("primitive type must be a string");
    static const std::pair<const char *, int> kTypes[] = {
        {"sphere", RM_SPHERE}, {"cube", RM_CUBE}, {"cylinder", RM_CYLINDER}, {"cone", RM_CONE},
        {"octahedron", RM_OCTAHEDRON}, {"torus", RM_TORUS}, {"capsule", RM_CAPSULE}, {"deathstar", RM_DEATHSTAR},
        {"rectangle", RM_RECTANGLE}, {"mandelbrot", RM_MANDELBROT}, {"mandelbulb", RM_MANDELBULB},
        {"mengersponge", RM_MENGERSPONGE}, {"sierpinski", RM_SIERPINSKI}, {"custom", RM_CUSTOM}};
    Primitive pr;
    bool known = false;
    for (auto &kt : kTypes)
      if (t->str == kt.first) { pr.type = kt.second; known = true; }
    if (!known) return fail("unknown primitive type \"" + t->str + "\"");
    if (p.has("ambient") && !vec(p, "ambient", 3, pr.amb, "primitive")) return false;
    if (p.has("diffuse") && !vec(p, "diffuse", 3, pr.dif, "primitive")) return false;
    if (p.has("specular") && !vec(p, "specular", 3, pr.spec, "primitive")) return false;
    if (p.has("reflective") && !vec(p, "reflective", 3, pr.refl, "primitive")) return false;
    if (p.has("transparent") && !vec(p, "transparent", 3, pr.transp, "primitive")) return false;
    double v;
    if (p.has("shininess")) { if (!num(p, "shininess", &v, "primitive")) return false; pr.shininess = (float)v; }
    if (p.has("ior")) { if (!num(p, "ior", &v, "primitive")) return false; pr.ior = (float)v; }
    if (p.has("blend")) { if (!num(p, "blend", &v, "primitive")) return false; pr.blend = (float)v; }
    if (p.has("textureFile")) {
      const JValue *tf = p.find("textureFile");
      if (tf->type != JValue::String) return fail("primitive textureFile must be a string");
      pr.textured = true;
      pr.texFile = baseDir.empty() ? tf->str : (baseDir + "/" + tf->str);
      const JValue *tu = p.find("textureU"), *tv = p.find("textureV");
      pr.repeatU = (tu && tu->isNumber()) ? (float)tu->num : 1.0f;
      pr.repeatV = (tv && tv->isNumber()) ? (float)tv->num : 1.0f;
    }
    if (p.has("bumpMapFile") && p.find("bumpMapFile")->type != JValue::String)
      return fail("primitive bumpMapFile must be a string");
    node->prims.push_back(pr);
    return true;
  }

  // scenefilereader.cpp:664-856
  bool groupData(const JValue &g, Node *node) {
    if (!onlyKeys(g, "group object", {"name", "translate", "rotate", "scale", "matrix", "lights", "primitives", "groups"}))
      return false;
    if (g.has("translate")) {
      if (!vec(g, "translate", 3, node->translate, "group")) return false;
      node->hasT = true;
    }
    if (g.has("rotate")) {
      float r[4];
      if (!vec(g, "rotate", 4, r, "group")) return false;
      node->rotAxis[0] = r[0]; node->rotAxis[1] = r[1]; node->rotAxis[2] = r[2];
      node->rotAngle = (float)((double)g.find("rotate")->arr[3].num * kPi / 180.0);
      node->hasR = true;
    }
    if (g.has("scale")) {
      if (!vec(g, "scale", 3, node->scale, "group")) return false;
      node->hasS = true;
    }
    if (g.has("matrix")) {
      const JValue *m = g.find("matrix");
      if (m->type != JValue::Array || m->arr.size() != 4) return fail("group matrix must be 4x4");
      for (int r = 0; r < 4; r++) {
        const JValue &row = m->arr[r];
        if (row.type != JValue::Array || row.arr.size() != 4) return fail("group matrix must be 4x4");
        for (int c = 0; c < 4; c++) {
          if (!row.arr[c].isNumber()) return fail("group matrix must contain numbers");
          node->matrix.m[c * 4 + r] = (float)row.arr[c].num;  // JSON is row-major
        }
      }
      node->hasM = true;
    }
    if (g.has("lights")) {
      const JValue *ls = g.find("lights");
      if (ls->type != JValue::Array) return fail("group lights must be an array");
      for (auto &l : ls->arr) {
        if (l.type != JValue::Object) return fail("light must be an object");
        if (!light(l, node)) return false;
      }
    }
    if (g.has("primitives")) {
      const JValue *ps = g.find("primitives");
      if (ps->type != JValue::Array) return fail("group primitives must be an array");
      for (auto &p : ps->arr) {
        if (p.type != JValue::Object) return fail("primitive must be an object");
        if (!primitive(p, node)) return false;
      }
    }
    if (g.has("groups") && !groups(*g.find("groups"), node)) return false;
    return true;
  }

  // scenefilereader.cpp:858-887: a group whose name matches a template IS that template node.
  bool groups(const JValue &gs, Node *parent) {
    if (gs.type != JValue::Array) return fail("groups must be an array");
    for (auto &g : gs.arr) {
      if (g.type != JValue::Object) return fail("group items must be of type object");
      if (const JValue *n = g.find("name")) {
        if (n->type != JValue::String) return fail("group name must be a string");
        auto it = templates.find(n->str);
        if (it != templates.end()) { parent->children.push_back(it->second); continue; }
      }
      nodes.emplace_back(new Node);
      Node *node = nodes.back().get();
      parent->children.push_back(node);
      if (!groupData(g, node)) return false;
    }
    return true;
  }

  // scenefilereader.cpp:604-662
  bool templateGroups(const JValue &ts) {
    if (ts.type != JValue::Array) return fail("templateGroups must be an array");
    for (auto &t : ts.arr) {
      if (t.type != JValue::Object) return fail("templateGroup items must be of type object");
      const JValue *n = t.find("name");
      if (!n) return fail("missing required field \"name\" on templateGroup object");
      if (n->type != JValue::String) return fail("templateGroup name must be a string");
      nodes.emplace_back(new Node);
      Node *node = nodes.back().get();
      templates[n->str] = node;
      if (!groupData(t, node)) return false;
    }
    return true;
  }

  // sceneparser.cpp:38-108: ctm = parent · M · T · R · S, accumulated scale = parentScale · S.
  // `path` holds the nodes between the root and n: a template group that (directly or not) contains itself is cut where it
  // re-enters instead of being expanded 2^depth times; flattenBudget bounds the total expansion of hostile files.
  std::vector<const Node *> path;
  long flattenBudget = 1 << 20;
  void flatten(const Node *n, const M4 &parent, const M4 &parentScale, RmScene &sc, std::vector<M4> &lightCtms, int depth) {
    if (depth > 64 || --flattenBudget < 0) return;
    for (const Node *p : path)
      if (p == n) return;  // cyclic template reference
    path.push_back(n);
    M4 T = identity(), R = identity(), S = identity(), Mx = identity();
    if (n->hasM) Mx = n->matrix;
    if (n->hasS) { S.m[0] = n->scale[0]; S.m[5] = n->scale[1]; S.m[10] = n->scale[2]; }
    if (n->hasR && !(n->rotAxis[0] == 0.0f && n->rotAxis[1] == 0.0f && n->rotAxis[2] == 0.0f))
      R = rotation(n->rotAngle, n->rotAxis);
    if (n->hasT) { T.m[12] = n->translate[0]; T.m[13] = n->translate[1]; T.m[14] = n->translate[2]; }
    const M4 ctm = mul(mul(mul(mul(parent, Mx), T), R), S);
    const M4 accS = mul(parentScale, S);
    for (const Primitive &p : n->prims) {
      RmObject o{};
      o.type = p.type;
      const M4 inv = inverse(ctm);  // RayMarchObj::m_ctmInv, raymarchobj.h:13
      std::memcpy(o.invModel, inv.m, sizeof(inv.m));
      o.scaleFactor = std::fmin(accS.m[0], std::fmin(accS.m[5], accS.m[10]));  // realtimerender.cpp:749-751
      o.shininess = p.shininess; o.blend = p.blend; o.ior = p.ior;
      for (int i = 0; i < 3; i++) {
        o.cAmbient[i] = p.amb[i]; o.cDiffuse[i] = p.dif[i]; o.cSpecular[i] = p.spec[i];
        o.cReflective[i] = p.refl[i]; o.cTransparent[i] = p.transp[i];
      }
      o.texLoc = -1;
      o.repeatU = p.repeatU; o.repeatV = p.repeatV;
      o.isEmissive = 0; o.lightIdx = -1;
      sc.objects.push_back(o);
      sc.textures.push_back(p.textured ? p.texFile : std::string());
    }
    for (const Light &l : n->lights) {  // sceneparser.cpp:15-31
      RmLight out{};
      out.type = l.type;
      const float origin[4] = {0, 0, 0, 1};
      float pos[4], dir[4];
      mulVec(ctm, origin, pos);
      mulVec(ctm, l.dir, dir);
      for (int i = 0; i < 3; i++) { out.color[i] = l.color[i]; out.pos[i] = pos[i]; out.dir[i] = dir[i]; out.func[i] = l.func[i]; }
      out.angle = l.angle; out.penumbra = l.penumbra;
      out.intensity = 0.0f;  // dropped by the aggregate initialiser, sceneparser.cpp:18-30
      out.twoSided = (l.type == RM_LIGHT_AREA) ? 1 : 0;
      if (l.type == RM_LIGHT_AREA) {
        // configureLightsUniforms, realtimerender.cpp:682-693: points[k] = ctm · corner k of the unit square in the
        // light's plane (realtime.h:136-141: tl, tr, br, bl) — the rectangle the LTC integral runs over
        static const float kCorners[4][4] = {{-0.5f, 0.5f, 0.0f, 1.0f}, {0.5f, 0.5f, 0.0f, 1.0f}, {0.5f, -0.5f, 0.0f, 1.0f}, {-0.5f, -0.5f, 0.0f, 1.0f}};
        for (int k = 0; k < 4; k++) {
          float w[4];
          mulVec(ctm, kCorners[k], w);
          for (int i = 0; i < 3; i++) out.points[k][i] = w[i];
        }
      }
      sc.lights.push_back(out);
      lightCtms.push_back(ctm);
    }
    for (const Node *c : n->children) flatten(c, ctm, accS, sc, lightCtms, depth + 1);
    path.pop_back();
  }

  bool read(const std::string &text, RmScene &sc) {
    JParser jp(text);
    JValue doc;
    if (!jp.document(doc)) return fail("could not parse JSON: " + jp.err);
    if (doc.type != JValue::Object) return fail("document is not an object");
    if (!doc.has("globalData")) return fail("missing required field \"globalData\" on root object");
    if (!doc.has("cameraData")) return fail("missing required field \"cameraData\" on root object");
    if (!onlyKeys(doc, "root object", {"globalData", "cameraData", "name", "groups", "templateGroups"})) return false;
    const JValue *g = doc.find("globalData"), *c = doc.find("cameraData");
    // QJsonValue::toObject() of a non-object is an empty object → the required-field checks fire
    if (!globalData(*g, sc)) return fail("could not parse \"globalData\"");
    if (!cameraData(*c, sc)) return fail("could not parse \"cameraData\"");
    if (doc.has("templateGroups") && !templateGroups(*doc.find("templateGroups"))) return false;
    Node root;
    if (doc.has("groups") && !groups(*doc.find("groups"), &root)) return false;
    std::vector<M4> lightCtms;
    flatten(&root, identity(), identity(), sc, lightCtms, 0);
    // RayMarchScene::initScene, raymarchscene.cpp:121-133: one emissive RECTANGLE per area light
    for (size_t i = 0; i < sc.lights.size(); i++) {
      if (sc.lights[i].type != RM_LIGHT_AREA) continue;
      RmObject o{};
      o.type = RM_RECTANGLE;
      const M4 inv = inverse(lightCtms[i]);
      std::memcpy(o.invModel, inv.m, sizeof(inv.m));
      o.scaleFactor = 1.0f;
      o.texLoc = -1;
      o.isEmissive = 1;
      for (int k = 0; k < 3; k++) o.color[k] = sc.lights[i].color[k];
      o.lightIdx = (int)i;
      sc.objects.push_back(o);
      sc.textures.push_back(std::string());
    }
    // texture slots in first-use order (configureShapesUniforms, realtimerender.cpp:735-806)
    std::map<std::string, int> slot;
    for (size_t i = 0; i < sc.objects.size(); i++) {
      if (sc.textures[i].empty()) continue;
      auto it = slot.find(sc.textures[i]);
      if (it == slot.end()) it = slot.emplace(sc.textures[i], (int)slot.size()).first;
      sc.objects[i].texLoc = it->second;
    }
    return true;
  }
};

int load_text(const std::string &text, const std::string &baseDir, RmScene **out) {
  if (!out) { set_error("null output pointer"); return RM_ERR_INVALID_ARGUMENT; }
  *out = nullptr;
  std::unique_ptr<RmScene> sc(new RmScene);
  Reader rd;
  rd.baseDir = baseDir;
  if (!rd.read(text, *sc)) { set_error(rd.err); return RM_ERR_PARSE; }
  *out = sc.release();
  return RM_OK;
}

std::string parentDir(const std::string &p) {
  size_t i = p.find_last_of('/');
  if (i == std::string::npos) return std::string();
  return p.substr(0, i);
}

}  // namespace
}  // namespace rm

extern "C" {

int rm_scene_load(const char *path, RmScene **out) {
  if (!path) { set_error("null path"); return RM_ERR_INVALID_ARGUMENT; }
  std::ifstream f(path, std::ios::binary);
  if (!f) { set_error(std::string("could not open ") + path); return RM_ERR_IO; }
  std::stringstream ss;
  ss << f.rdbuf();
  return load_text(ss.str(), parentDir(parentDir(path)), out);
}
int rm_scene_load_string(const char *json, RmScene **out) {
  if (!json) { set_error("null json"); return RM_ERR_INVALID_ARGUMENT; }
  return load_text(json, std::string(), out);
}
void rm_scene_free(RmScene *scene) { delete scene; }
int rm_scene_num_objects(const RmScene *scene) { return scene ? (int)scene->objects.size() : -1; }
int rm_scene_num_lights(const RmScene *scene) { return scene ? (int)scene->lights.size() : -1; }
const RmObject *rm_scene_objects(const RmScene *scene) { return (scene && !scene->objects.empty()) ? scene->objects.data() : nullptr; }
const RmLight *rm_scene_lights(const RmScene *scene) { return (scene && !scene->lights.empty()) ? scene->lights.data() : nullptr; }
int rm_scene_globals(const RmScene *scene, const RmHostSettings *hs, RmGlobals *out) {
  if (!scene || !out) { set_error("null argument"); return RM_ERR_INVALID_ARGUMENT; }
  std::memset(out, 0, sizeof(*out));
  out->ka = scene->ka; out->kd = scene->kd; out->ks = scene->ks; out->kt = scene->kt;
  out->power = hs ? hs->power : 8.0f;
  out->juliaSeed[0] = hs ? hs->juliaSeed[0] : 0.0f;
  out->juliaSeed[1] = hs ? hs->juliaSeed[1] : 0.0f;
  out->iTime = 0.0f;
  out->isTwoD = hs ? hs->twoDSpace : 0;
  return RM_OK;
}
int rm_scene_camera_data(const RmScene *scene, RmCameraData *out) {
  if (!scene || !out) { set_error("null argument"); return RM_ERR_INVALID_ARGUMENT; }
  *out = scene->cam;
  return RM_OK;
}
const char *rm_scene_object_texture(const RmScene *scene, int i) {
  if (!scene || i < 0 || i >= (int)scene->textures.size() || scene->textures[i].empty()) return nullptr;
  return scene->textures[i].c_str();
}

int rm_abi_sizeof(int which) {
  switch (which) {
    case 0: return (int)sizeof(RmObject);
    case 1: return (int)sizeof(RmLight);
    case 2: return (int)sizeof(RmCamera);
    case 3: return (int)sizeof(RmGlobals);
    case 4: return (int)sizeof(RmSettings);
    case 5: return (int)sizeof(RmCounters);
    case 6: return (int)sizeof(RmHostSettings);
    case 7: return (int)sizeof(RmCameraData);
    case 8: return (int)sizeof(RmTexture);
    case 9: return (int)sizeof(RmPostSettings);
    case 10: return (int)sizeof(RmResources);
    default: return -1;
  }
}

}  // extern "C"
