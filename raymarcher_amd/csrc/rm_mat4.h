// rm_mat4.h — 4×4 column-major binary32 matrices for the host side (camera, scene-graph flattening): the evaluation
// orders are those of the reference's math library (vendored glm: mat4 * mat4 by column combination left to right,
// inverse by cofactors with one reciprocal of the determinant), so loader and camera tables agree with the reference's
// own to the bit wherever the inputs do (tests/test_host_loader.py).
#pragma once

namespace rm {

// element (row r, col c) at m[c*4 + r]
struct Mat4 {
  float m[16];
  float &at(int r, int c) { return m[c * 4 + r]; }
  float at(int r, int c) const { return m[c * 4 + r]; }
};
inline Mat4 mat_identity() {
  Mat4 r{};
  for (int i = 0; i < 4; i++) r.at(i, i) = 1.0f;
  return r;
}
// column-combination product: each result column is Σ_k A.col(k)·B(k,c), accumulated left to right in
// binary32 (the evaluation order of the reference's math library, so camera matrices agree to the bit
// wherever the inputs do).
inline Mat4 mat_mul(const Mat4 &A, const Mat4 &B) {
  Mat4 R{};
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 4; r++) {
      float acc = A.at(r, 0) * B.at(0, c);
      acc = acc + A.at(r, 1) * B.at(1, c);
      acc = acc + A.at(r, 2) * B.at(2, c);
      acc = acc + A.at(r, 3) * B.at(3, c);
      R.at(r, c) = acc;
    }
  return R;
}
// Cofactor (adjugate / determinant) inverse in binary32.
inline Mat4 mat_inverse(const Mat4 &M) {
  auto m = [&](int c, int r) { return M.m[c * 4 + r]; };  // m(col,row)
  // 2×2 sub-determinants of rows/cols of the lower-right 3×3 blocks
  float c00 = m(2, 2) * m(3, 3) - m(3, 2) * m(2, 3), c02 = m(1, 2) * m(3, 3) - m(3, 2) * m(1, 3),
        c03 = m(1, 2) * m(2, 3) - m(2, 2) * m(1, 3);
  float c04 = m(2, 1) * m(3, 3) - m(3, 1) * m(2, 3), c06 = m(1, 1) * m(3, 3) - m(3, 1) * m(1, 3),
        c07 = m(1, 1) * m(2, 3) - m(2, 1) * m(1, 3);
  float c08 = m(2, 1) * m(3, 2) - m(3, 1) * m(2, 2), c10 = m(1, 1) * m(3, 2) - m(3, 1) * m(1, 2),
        c11 = m(1, 1) * m(2, 2) - m(2, 1) * m(1, 2);
  float c12 = m(2, 0) * m(3, 3) - m(3, 0) * m(2, 3), c14 = m(1, 0) * m(3, 3) - m(3, 0) * m(1, 3),
        c15 = m(1, 0) * m(2, 3) - m(2, 0) * m(1, 3);
  float c16 = m(2, 0) * m(3, 2) - m(3, 0) * m(2, 2), c18 = m(1, 0) * m(3, 2) - m(3, 0) * m(1, 2),
        c19 = m(1, 0) * m(2, 2) - m(2, 0) * m(1, 2);
  float c20 = m(2, 0) * m(3, 1) - m(3, 0) * m(2, 1), c22 = m(1, 0) * m(3, 1) - m(3, 0) * m(1, 1),
        c23 = m(1, 0) * m(2, 1) - m(2, 0) * m(1, 1);
  const float f0[4] = {c00, c00, c02, c03}, f1[4] = {c04, c04, c06, c07}, f2[4] = {c08, c08, c10, c11};
  const float f3[4] = {c12, c12, c14, c15}, f4[4] = {c16, c16, c18, c19}, f5[4] = {c20, c20, c22, c23};
  const float v0[4] = {m(1, 0), m(0, 0), m(0, 0), m(0, 0)}, v1[4] = {m(1, 1), m(0, 1), m(0, 1), m(0, 1)};
  const float v2[4] = {m(1, 2), m(0, 2), m(0, 2), m(0, 2)}, v3[4] = {m(1, 3), m(0, 3), m(0, 3), m(0, 3)};
  Mat4 inv{};
  for (int i = 0; i < 4; i++) {
    const float sa = (i & 1) ? -1.0f : 1.0f, sb = -sa;
    inv.m[0 * 4 + i] = ((v1[i] * f0[i] - v2[i] * f1[i]) + v3[i] * f2[i]) * sa;
    inv.m[1 * 4 + i] = ((v0[i] * f0[i] - v2[i] * f3[i]) + v3[i] * f4[i]) * sb;
    inv.m[2 * 4 + i] = ((v0[i] * f1[i] - v1[i] * f3[i]) + v3[i] * f5[i]) * sa;
    inv.m[3 * 4 + i] = ((v0[i] * f2[i] - v1[i] * f4[i]) + v2[i] * f5[i]) * sb;
  }
  const float d0 = m(0, 0) * inv.m[0], d1 = m(0, 1) * inv.m[4], d2 = m(0, 2) * inv.m[8], d3 = m(0, 3) * inv.m[12];
  const float ood = 1.0f / ((d0 + d1) + (d2 + d3));
  for (float &x : inv.m) x = x * ood;
  return inv;
}

inline void mat_mul_vec(const Mat4 &A, const float v[4], float out[4]) {
  for (int r = 0; r < 4; r++) {
    float acc = A.at(r, 0) * v[0];
    acc = acc + A.at(r, 1) * v[1];
    acc = acc + A.at(r, 2) * v[2];
    acc = acc + A.at(r, 3) * v[3];
    out[r] = acc;
  }
}

}  // namespace rm
