// oracle/ref/dump_tables.cpp — TEST INFRASTRUCTURE, container-only.
// Driver around the reference's OWN, UNMODIFIED loader + camera (compiled from where they lie under
// /root/reference by oracle/ref/Makefile): prints, as JSON, the uniform tables Realtime::configure*Uniforms
// would upload for one scenefile (src/realtimerender.cpp:596-811).  Used only to generate
// tests/golden/host_tables.json; nothing here ships or runs on the GPU box.
#include <cstdio>
#include <string>

#include "camera/camera.h"
#include "settings.h"
#include "utils/ltc_matrix.h"
#include "utils/sceneparser.h"

Settings settings;  // the reference declares it extern (src/settings.h:55)

static void mat(const char *name, const glm::mat4 &m, bool comma = true) {
  std::printf("\"%s\": [", name);
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 4; r++) std::printf("%s%.9g", (c || r) ? ", " : "", m[c][r]);
  std::printf("]%s", comma ? ", " : "");
}
static void vec(const char *name, const glm::vec4 &v, int n, bool comma = true) {
  std::printf("\"%s\": [", name);
  for (int i = 0; i < n; i++) std::printf("%s%.9g", i ? ", " : "", v[i]);
  std::printf("]%s", comma ? ", " : "");
}

int main(int argc, char **argv) {
  // --ltc FILE: the two 64×64 RGBA float tables Realtime::loadMTexture / loadLTUTexture hand to glTexImage2D
  // (realtimerender.cpp:896-930), raw binary32, LTC1 then LTC2 — for the fixture generator to quantise as GL_RGBA8 does
  if (argc == 3 && std::string(argv[1]) == "--ltc") {
    FILE *f = std::fopen(argv[2], "wb");
    if (!f) return 3;
    static_assert(sizeof(LTC1) == 64 * 64 * 4 * sizeof(float) && sizeof(LTC2) == sizeof(LTC1), "64x64 RGBA");
    std::fwrite(LTC1, 1, sizeof(LTC1), f);
    std::fwrite(LTC2, 1, sizeof(LTC2), f);
    std::fclose(f);
    return 0;
  }
  if (argc < 4) return 2;
  Settings s;
  s.sceneFilePath = argv[1];
  s.screenWidth = std::atoi(argv[2]);
  s.screenHeight = std::atoi(argv[3]);
  s.nearPlane = 0.1f;
  s.farPlane = 100.f;
  RenderData rd;
  std::fflush(stdout);
  FILE *saved = stdout;
  (void)saved;
  bool ok = SceneParser::parse(s.sceneFilePath, rd);
  if (!ok) { std::printf("\n@@JSON {\"ok\": false}\n"); return 0; }
  Camera cam;
  cam.initializeCamera(rd.cameraData, s);
  glm::mat4 view = cam.getViewMatrix(), proj = cam.getProjMatrix();
  glm::mat4 inv = glm::inverse(proj * view);
  std::printf("\n@@JSON {\"ok\": true, ");
  std::printf("\"ka\": %.9g, \"kd\": %.9g, \"ks\": %.9g, \"kt\": %.9g, ", rd.globalData.ka, rd.globalData.kd, rd.globalData.ks,
              rd.globalData.kt);
  mat("view", view); mat("proj", proj); mat("invProjView", inv);
  vec("camPos", rd.cameraData.pos, 4); vec("camLook", rd.cameraData.look, 4); vec("camUp", rd.cameraData.up, 4);
  std::printf("\"heightAngle\": %.9g, ", rd.cameraData.heightAngle);
  std::printf("\"objects\": [");
  for (size_t i = 0; i < rd.shapes.size(); i++) {
    const RenderShapeData &sh = rd.shapes[i];
    const SceneMaterial &m = sh.primitive.material;
    std::printf("%s{\"type\": %d, ", i ? ", " : "", (int)sh.primitive.type);
    mat("ctm", sh.ctm); mat("invModel", glm::inverse(sh.ctm));
    float sf = fmin(sh.scale[0][0], fmin(sh.scale[1][1], sh.scale[2][2]));
    std::printf("\"scaleFactor\": %.9g, \"shininess\": %.9g, \"blend\": %.9g, \"ior\": %.9g, ", sf, m.shininess, m.blend, m.ior);
    vec("cAmbient", m.cAmbient, 3); vec("cDiffuse", m.cDiffuse, 3); vec("cSpecular", m.cSpecular, 3);
    vec("cReflective", m.cReflective, 3); vec("cTransparent", m.cTransparent, 3);
    std::printf("\"textured\": %s, \"textureFile\": \"%s\", \"repeatU\": %.9g, \"repeatV\": %.9g}", m.textureMap.isUsed ? "true" : "false",
                m.textureMap.isUsed ? m.textureMap.filename.c_str() : "",
                m.textureMap.isUsed ? m.textureMap.repeatU : 0.f, m.textureMap.isUsed ? m.textureMap.repeatV : 0.f);
  }
  std::printf("], \"lights\": [");
  for (size_t i = 0; i < rd.lights.size(); i++) {
    const SceneLightData &l = rd.lights[i];
    std::printf("%s{\"type\": %d, ", i ? ", " : "", (int)l.type);
    vec("color", l.color, 3); vec("pos", l.pos, 3); vec("dir", l.dir, 3);
    vec("func", glm::vec4(l.function, 0.f), 3);
    std::printf("\"angle\": %.9g, \"penumbra\": %.9g", l.angle, l.penumbra);
    if (l.type == LightType::LIGHT_AREA) {  // what the emissive rectangle and the `points` uniforms are built from
      std::printf(", \"width\": %.9g, \"height\": %.9g, \"intensity\": %.9g, ", l.width, l.height, l.intensity);
      mat("ctm", l.ctm); mat("ctmInv", glm::inverse(l.ctm), false);
    }
    std::printf("}");
  }
  std::printf("]}\n");
  return 0;
}
