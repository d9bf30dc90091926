// grx_eng_narrowphase.h -- K8 narrow phase, analytic pairs: plane / sphere / capsule / box combinations, contact frames, the contact list.
// A FRAGMENT of csrc/grx_engine.h: textually included INSIDE `template <class S> struct GrxEngine { ... }` (every function here is a static member), in the order the engine
// header lists; not a standalone header.  The split is purely textual (round 5): the token stream of the translation units is unchanged.
// ------------------------------------------------------------------------------------------
// K8 collision: static candidate list -> narrow phase
// ------------------------------------------------------------------------------------------
GRX_MEM void grx_make_frame(float* f) {
  float* x = f; float* y = f + 3; float* z = f + 6;
  if (x[1] < 0.5f && x[1] > -0.5f) { y[0] = 0; y[1] = 1; y[2] = 0; } else { y[0] = 0; y[1] = 0; y[2] = 1; }
  float d = dot3f(x, y); y[0] -= d * x[0]; y[1] -= d * x[1]; y[2] -= d * x[2];
  float n = 1.0f / sqrtf(dot3f(y, y)); y[0] *= n; y[1] *= n; y[2] *= n;
  cross3f(z, x, y);
}

// contact append: slot from an LDS counter; only the normal is stored here, the tangent frame is completed by
// grx_make_constraint (one lane per contact)
GRX_MEM void grx_add_contact(GrxCtx* c, int pair, const float* pos, const float* normal, float dist) {
  int slot = GRX_ATOMIC_ADD(&c->cnt[0], 1);
  if (slot >= c->maxcon) { c->cnt[2] |= GRX_ST_CON_OVERFLOW; return; }
  c->con_dist[slot] = dist; c->con_pair[slot] = pair;
  for (int k = 0; k < 3; k++) { c->con_pos[3 * slot + k] = pos[k]; c->con_frame[3 * slot + k] = normal[k]; }
}

GRX_MEM void grx_plane_box(const GrxModel* m, GrxCtx* c, int pair, int g1, int g2, float margin) {
  const float* pp = c->gxpos + 3 * g1; const float* pm = c->gxmat + 9 * g1;
  const float* bp = c->gxpos + 3 * g2; const float* bm = c->gxmat + 9 * g2; const float* sz = m->geom_size + 3 * g2;
  float n[3] = {pm[2], pm[5], pm[8]};
  float sx = sz[0], sy = sz[1], szz = sz[2];
  int cnt = 0;
  for (int k = 0; k < 8; k++) {
    float loc[3] = {(k & 1) ? sx : -sx, (k & 2) ? sy : -sy, (k & 4) ? szz : -szz}, w[3];
    mulMatVec3f(w, bm, loc); w[0] += bp[0]; w[1] += bp[1]; w[2] += bp[2];
    float d[3] = {w[0] - pp[0], w[1] - pp[1], w[2] - pp[2]};
    float dist = dot3f(d, n);
    if (dist > margin || cnt >= 4) continue;
    float pos[3] = {w[0] - 0.5f * dist * n[0], w[1] - 0.5f * dist * n[1], w[2] - 0.5f * dist * n[2]};
    grx_add_contact(c, pair, pos, n, dist); cnt++;
  }
}

#define GRX_SEL3(a0, a1, a2, i) ((i) == 0 ? (a0) : ((i) == 1 ? (a1) : (a2)))
#define GRX_SEL6(v, i) ((i) == 0 ? (v)[0] : ((i) == 1 ? (v)[1] : ((i) == 2 ? (v)[2] : ((i) == 3 ? (v)[3] : ((i) == 4 ? (v)[4] : (v)[5])))))

GRX_MEM void grx_plane_sphere(const GrxModel* m, GrxCtx* c, int pair, int g1, int g2, float margin) {
  float n[3] = {c->gxmat[9 * g1 + 2], c->gxmat[9 * g1 + 5], c->gxmat[9 * g1 + 8]};
  const float* ce = c->gxpos + 3 * g2; float r = m->geom_size[3 * g2];
  float d[3] = {ce[0] - c->gxpos[3 * g1], ce[1] - c->gxpos[3 * g1 + 1], ce[2] - c->gxpos[3 * g1 + 2]};
  float dist = dot3f(d, n) - r;
  if (dist > margin) return;
  float pos[3] = {ce[0] - n[0] * (r + 0.5f * dist), ce[1] - n[1] * (r + 0.5f * dist), ce[2] - n[2] * (r + 0.5f * dist)};
  grx_add_contact(c, pair, pos, n, dist);
}

// sphere (geom1) vs box (geom2): closest point of the box to the sphere centre; normal from the sphere to the box
GRX_MEM void grx_sphere_box(const GrxModel* m, GrxCtx* c, int pair, int g1, int g2, float margin) {
  const float* ce = c->gxpos + 3 * g1; float r = m->geom_size[3 * g1];
  const float* bp = c->gxpos + 3 * g2; const float* bm = c->gxmat + 9 * g2; const float* sz = m->geom_size + 3 * g2;
  float dw[3] = {ce[0] - bp[0], ce[1] - bp[1], ce[2] - bp[2]}, loc[3];
  mulMatTVec3f(loc, bm, dw);
  float s0 = sz[0], s1 = sz[1], s2 = sz[2];
  float c0 = fminf(s0, fmaxf(-s0, loc[0])), c1 = fminf(s1, fmaxf(-s1, loc[1])), c2 = fminf(s2, fmaxf(-s2, loc[2]));
  float nl[3], dist;
  if (c0 != loc[0] || c1 != loc[1] || c2 != loc[2]) {
    float dv[3] = {c0 - loc[0], c1 - loc[1], c2 - loc[2]};
    float len = sqrtf(dot3f(dv, dv));
    dist = len - r;
    if (dist > margin) return;
    float li = 1.0f / len; nl[0] = dv[0] * li; nl[1] = dv[1] * li; nl[2] = dv[2] * li;
  } else {
    float d0 = s0 - fabsf(loc[0]), d1 = s1 - fabsf(loc[1]), d2 = s2 - fabsf(loc[2]);
    int ax = 0; float best = d0;
    if (d1 < best) { best = d1; ax = 1; }
    if (d2 < best) { best = d2; ax = 2; }
    float sg = (GRX_SEL3(loc[0], loc[1], loc[2], ax) >= 0) ? -1.0f : 1.0f;
    nl[0] = (ax == 0) ? sg : 0.0f; nl[1] = (ax == 1) ? sg : 0.0f; nl[2] = (ax == 2) ? sg : 0.0f;
    dist = -best - r;
  }
  float n[3]; mulMatVec3f(n, bm, nl);
  float pos[3] = {ce[0] + n[0] * (r + 0.5f * dist), ce[1] + n[1] * (r + 0.5f * dist), ce[2] + n[2] * (r + 0.5f * dist)};
  grx_add_contact(c, pair, pos, n, dist);
}

// plane vs capsule: the two end spheres
GRX_MEM void grx_plane_capsule(const GrxModel* m, GrxCtx* c, int pair, int g1, int g2, float margin) {
  float n[3] = {c->gxmat[9 * g1 + 2], c->gxmat[9 * g1 + 5], c->gxmat[9 * g1 + 8]};
  const float* ce = c->gxpos + 3 * g2; const float* R = c->gxmat + 9 * g2;
  float r = m->geom_size[3 * g2], hl = m->geom_size[3 * g2 + 1], ax[3] = {R[2], R[5], R[8]};
  for (int e = -1; e <= 1; e += 2) {
    float p[3] = {ce[0] + e * hl * ax[0], ce[1] + e * hl * ax[1], ce[2] + e * hl * ax[2]};
    float d[3] = {p[0] - c->gxpos[3 * g1], p[1] - c->gxpos[3 * g1 + 1], p[2] - c->gxpos[3 * g1 + 2]};
    float dist = dot3f(d, n) - r;
    if (dist > margin) continue;
    float pos[3] = {p[0] - n[0] * (r + 0.5f * dist), p[1] - n[1] * (r + 0.5f * dist), p[2] - n[2] * (r + 0.5f * dist)};
    grx_add_contact(c, pair, pos, n, dist);
  }
}

GRX_MEM float grx_box_point_dist2(float s0, float s1, float s2, float p0, float p1, float p2) {
  float d0 = p0 - fminf(s0, fmaxf(-s0, p0)), d1 = p1 - fminf(s1, fmaxf(-s1, p1)), d2 = p2 - fminf(s2, fmaxf(-s2, p2));
  return d0 * d0 + d1 * d1 + d2 * d2;
}
// sphere of radius r at box-frame point p against the box (normal from the sphere to the box); returns 1 if a contact was made
GRX_MEM int grx_sphere_box_local(GrxCtx* c, int pair, const float* bp, const float* bm, float s0, float s1, float s2, const float* p, float r, float margin) {
  float c0 = fminf(s0, fmaxf(-s0, p[0])), c1 = fminf(s1, fmaxf(-s1, p[1])), c2 = fminf(s2, fmaxf(-s2, p[2]));
  float nl[3], dist;
  if (c0 != p[0] || c1 != p[1] || c2 != p[2]) {
    float dv[3] = {c0 - p[0], c1 - p[1], c2 - p[2]};
    float len = sqrtf(dot3f(dv, dv));
    dist = len - r;
    if (dist > margin) return 0;
    float li = 1.0f / len; nl[0] = dv[0] * li; nl[1] = dv[1] * li; nl[2] = dv[2] * li;
  } else {
    float d0 = s0 - fabsf(p[0]), d1 = s1 - fabsf(p[1]), d2 = s2 - fabsf(p[2]);
    int ax = 0; float best = d0;
    if (d1 < best) { best = d1; ax = 1; }
    if (d2 < best) { best = d2; ax = 2; }
    float sg = (GRX_SEL3(p[0], p[1], p[2], ax) >= 0) ? -1.0f : 1.0f;
    nl[0] = (ax == 0) ? sg : 0.0f; nl[1] = (ax == 1) ? sg : 0.0f; nl[2] = (ax == 2) ? sg : 0.0f;
    dist = -best - r;
  }
  float n[3], pw[3]; mulMatVec3f(n, bm, nl); mulMatVec3f(pw, bm, p);
  float pos[3] = {pw[0] + bp[0] + n[0] * (r + 0.5f * dist), pw[1] + bp[1] + n[1] * (r + 0.5f * dist), pw[2] + bp[2] + n[2] * (r + 0.5f * dist)};
  grx_add_contact(c, pair, pos, n, dist);
  return 1;
}
// capsule vs capsule: closest points of the two axis segments (clamped), then a sphere-sphere contact (see oracle/grx_oracle.c)
GRX_MEM void grx_capsule_capsule(const GrxModel* m, GrxCtx* c, int pair, int g1, int g2, float margin) {
  const float* c1 = c->gxpos + 3 * g1; const float* R1 = c->gxmat + 9 * g1; const float* c2 = c->gxpos + 3 * g2; const float* R2 = c->gxmat + 9 * g2;
  const float r1 = m->geom_size[3 * g1], h1 = m->geom_size[3 * g1 + 1], r2 = m->geom_size[3 * g2], h2 = m->geom_size[3 * g2 + 1];
  const float a1[3] = {R1[2], R1[5], R1[8]}, a2[3] = {R2[2], R2[5], R2[8]}, w[3] = {c1[0] - c2[0], c1[1] - c2[1], c1[2] - c2[2]};
  const float b = dot3f(a1, a2), d = dot3f(a1, w), e = dot3f(a2, w), den = 1.0f - b * b;
  float x1 = den > GRX_MINVAL ? (b * e - d) / den : 0.0f;
  x1 = fminf(h1, fmaxf(-h1, x1));
  float x2 = b * x1 + e;
  if (x2 > h2) { x2 = h2; x1 = fminf(h1, fmaxf(-h1, b * x2 - d)); }
  else if (x2 < -h2) { x2 = -h2; x1 = fminf(h1, fmaxf(-h1, b * x2 - d)); }
  float p1[3], n[3];
  for (int k = 0; k < 3; k++) { p1[k] = c1[k] + x1 * a1[k]; n[k] = c2[k] + x2 * a2[k] - p1[k]; }
  const float len = sqrtf(dot3f(n, n));
  if (len < GRX_MINVAL) { n[0] = 1; n[1] = n[2] = 0; } else { const float li = 1.0f / len; n[0] *= li; n[1] *= li; n[2] *= li; }
  const float dist = len - r1 - r2;
  if (dist > margin) return;
  float pos[3] = {p1[0] + n[0] * (r1 + 0.5f * dist), p1[1] + n[1] * (r1 + 0.5f * dist), p1[2] + n[2] * (r1 + 0.5f * dist)};
  grx_add_contact(c, pair, pos, n, dist);
}
// sphere vs sphere and sphere (geom1) vs capsule (geom2): the capsule contributes the point of its axis segment closest to the sphere centre
GRX_MEM void grx_sphere_sphere_raw(GrxCtx* c, int pair, const float* c1, float r1, const float* c2, float r2, float margin) {
  float n[3] = {c2[0] - c1[0], c2[1] - c1[1], c2[2] - c1[2]};
  const float len = sqrtf(dot3f(n, n)), dist = len - r1 - r2;
  if (dist > margin) return;
  if (len < GRX_MINVAL) { n[0] = 1; n[1] = n[2] = 0; } else { const float li = 1.0f / len; n[0] *= li; n[1] *= li; n[2] *= li; }
  float pos[3] = {c1[0] + n[0] * (r1 + 0.5f * dist), c1[1] + n[1] * (r1 + 0.5f * dist), c1[2] + n[2] * (r1 + 0.5f * dist)};
  grx_add_contact(c, pair, pos, n, dist);
}
GRX_MEM void grx_sphere_capsule(const GrxModel* m, GrxCtx* c, int pair, int g1, int g2, float margin) {
  const float* c1 = c->gxpos + 3 * g1; const float* c2 = c->gxpos + 3 * g2; const float* R2 = c->gxmat + 9 * g2;
  const float ax[3] = {R2[2], R2[5], R2[8]}, d[3] = {c1[0] - c2[0], c1[1] - c2[1], c1[2] - c2[2]};
  const float h = m->geom_size[3 * g2 + 1], x = fminf(h, fmaxf(-h, dot3f(ax, d)));
  const float p2[3] = {c2[0] + x * ax[0], c2[1] + x * ax[1], c2[2] + x * ax[2]};
  grx_sphere_sphere_raw(c, pair, c1, m->geom_size[3 * g1], p2, m->geom_size[3 * g2], margin);
}
