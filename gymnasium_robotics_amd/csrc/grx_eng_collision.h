// grx_eng_collision.h -- K8: hull pairs (wave-cooperative), plane-vs-convex analytic routines, box-box by lane octets, broad phase (sweep, skin lists, wall lattice) and the collision driver.
// A FRAGMENT of csrc/grx_engine.h: textually included INSIDE `template <class S> struct GrxEngine { ... }` (every function here is a static member), in the order the engine
// header lists; not a standalone header.  The split is purely textual (round 5): the token stream of the translation units is unchanged.
// ------------------------------------------------------------------------------------------
// Hull-vs-convex pairs (the convex hull of a mesh against a primitive or another hull: the Fetch arm / gripper / base links, assets/fetch/
// robot.xml:16-93).  MuJoCo sends them through the same general convex routine as the ellipsoid / cylinder pairs; here the pair is
// handled by the WHOLE wavefront: the portal refinement runs in wave-uniform control flow (every lane holds the same values) and the hull
// support function is a cooperative scan over the vertices (grx_mesh_support), because a lane-private walk over a hull in global memory
// is a chain of dependent loads (~25 us per support point).  Candidates are rare -- one persistent pair per Fetch world passes the
// bounding-box filter, a contact exists in 0.04 % of the substeps -- and a separating direction found by one substep is kept for the next
// ones (c->meshcache): re-checking it costs ONE support evaluation instead of the six or seven of a fresh portal search, and a direction
// that still separates the two inflated geoms proves that the routine would report "no contact".
// ------------------------------------------------------------------------------------------
// separating-axis test of the two geoms' oriented bounding boxes (geom_aabb), each grown by margin / 2 (the oracle's obb_overlap)
// (written out with named scalars: an array indexed by a loop variable would live in scratch memory)
GRX_MEM int grx_obb_overlap(const GrxModel* m, const GrxCtx* c, int g1, int g2, float margin) {
  const float* R1 = c->gxmat + 9 * g1; const float* R2 = c->gxmat + 9 * g2; const float* a1 = m->geom_aabb + 6 * g1; const float* a2 = m->geom_aabb + 6 * g2;
  const float hm = 0.5f * margin;
  const float a10 = a1[0], a11 = a1[1], a12 = a1[2], e10 = a1[3] + hm, e11 = a1[4] + hm, e12 = a1[5] + hm;
  const float a20 = a2[0], a21 = a2[1], a22 = a2[2], e20 = a2[3] + hm, e21 = a2[4] + hm, e22 = a2[5] + hm;
  const float r100 = R1[0], r101 = R1[1], r102 = R1[2], r110 = R1[3], r111 = R1[4], r112 = R1[5], r120 = R1[6], r121 = R1[7], r122 = R1[8];
  const float r200 = R2[0], r201 = R2[1], r202 = R2[2], r210 = R2[3], r211 = R2[4], r212 = R2[5], r220 = R2[6], r221 = R2[7], r222 = R2[8];
  // centre offset in world coordinates, then in the frames of box 1 (ta) and box 2 (tb)
  const float tx = (c->gxpos[3 * g2] + r200 * a20 + r201 * a21 + r202 * a22) - (c->gxpos[3 * g1] + r100 * a10 + r101 * a11 + r102 * a12);
  const float ty = (c->gxpos[3 * g2 + 1] + r210 * a20 + r211 * a21 + r212 * a22) - (c->gxpos[3 * g1 + 1] + r110 * a10 + r111 * a11 + r112 * a12);
  const float tz = (c->gxpos[3 * g2 + 2] + r220 * a20 + r221 * a21 + r222 * a22) - (c->gxpos[3 * g1 + 2] + r120 * a10 + r121 * a11 + r122 * a12);
  const float ta0 = tx * r100 + ty * r110 + tz * r120, ta1 = tx * r101 + ty * r111 + tz * r121, ta2 = tx * r102 + ty * r112 + tz * r122;
  const float tb0 = tx * r200 + ty * r210 + tz * r220, tb1 = tx * r201 + ty * r211 + tz * r221, tb2 = tx * r202 + ty * r212 + tz * r222;
  // C_ij = A_i . B_j (columns of the two frames)
#define GRX_OBB_C(i, j) const float C##i##j = r10##i * r20##j + r11##i * r21##j + r12##i * r22##j, Q##i##j = fabsf(C##i##j);
  GRX_OBB_C(0, 0) GRX_OBB_C(0, 1) GRX_OBB_C(0, 2) GRX_OBB_C(1, 0) GRX_OBB_C(1, 1) GRX_OBB_C(1, 2) GRX_OBB_C(2, 0) GRX_OBB_C(2, 1) GRX_OBB_C(2, 2)
#undef GRX_OBB_C
  if (fabsf(ta0) > e10 + e20 * Q00 + e21 * Q01 + e22 * Q02) return 0;
  if (fabsf(ta1) > e11 + e20 * Q10 + e21 * Q11 + e22 * Q12) return 0;
  if (fabsf(ta2) > e12 + e20 * Q20 + e21 * Q21 + e22 * Q22) return 0;
  if (fabsf(tb0) > e20 + e10 * Q00 + e11 * Q10 + e12 * Q20) return 0;
  if (fabsf(tb1) > e21 + e10 * Q01 + e11 * Q11 + e12 * Q21) return 0;
  if (fabsf(tb2) > e22 + e10 * Q02 + e11 * Q12 + e12 * Q22) return 0;
  // axis A_i x B_j (unnormalised on both sides of the test; nearly parallel edges are left to the face axes)
#define GRX_OBB_EDGE(i, i1, i2, j, j1, j2) \
  if (1.0f - C##i##j * C##i##j >= 1e-6f) { \
    const float tp_ = fabsf(ta##i2 * C##i1##j - ta##i1 * C##i2##j); \
    const float ra_ = e1##i1 * Q##i2##j + e1##i2 * Q##i1##j, rb_ = e2##j1 * Q##i##j2 + e2##j2 * Q##i##j1; \
    if (tp_ > (ra_ + rb_) * 1.0001f + 1e-7f) return 0; }
  GRX_OBB_EDGE(0, 1, 2, 0, 1, 2) GRX_OBB_EDGE(0, 1, 2, 1, 2, 0) GRX_OBB_EDGE(0, 1, 2, 2, 0, 1)
  GRX_OBB_EDGE(1, 2, 0, 0, 1, 2) GRX_OBB_EDGE(1, 2, 0, 1, 2, 0) GRX_OBB_EDGE(1, 2, 0, 2, 0, 1)
  GRX_OBB_EDGE(2, 0, 1, 0, 1, 2) GRX_OBB_EDGE(2, 0, 1, 1, 2, 0) GRX_OBB_EDGE(2, 0, 1, 2, 0, 1)
#undef GRX_OBB_EDGE
  return 1;
}

// Joint-box gate of a hull pair (mjcf/pair_gates.py): the two bodies are separated by at most three hinge / slide joints, and for joint values inside the
// gate's box the compiler has PROVEN the two margin-inflated geoms disjoint (rigorous distance bound on a grid + a Lipschitz bound in between).  1 = inside
// the box: the pair cannot produce a contact in this configuration and leaves the candidate sweep -- the Fetch arm's torso / shoulder pair, 1.9 cm apart in
// every pose the tasks reach, no longer walks through the bounding-box filter and the hull routine in every substep of every world.
GRX_MEM int grx_gate_clear(const GrxModel* m, const GrxCtx* c, int gi) {
  const int* qa = m->gate_qadr + 3 * gi; const float* bx = m->gate_box + 6 * gi;
  int ok = 1;
  for (int k = 0; k < 3; k++) { const int a = qa[k]; if (a >= 0) { const float q = c->qpos[a]; ok &= (q > bx[2 * k]) & (q < bx[2 * k + 1]); } }
  return ok;
}
// the queued hull-vs-convex pairs of this pass, one after the other, all lanes on each (wave-uniform code)
GRX_MEM void grx_mesh_pairs(const GrxModel* m, GrxCtx* c, const int* queue, int nq, int lane_) {
  for (int e = 0; e < nq; e++) {
    const int pair = queue[e], g1 = m->pair_geom1[pair], g2 = m->pair_geom2[pair];
    const float margin = m->pair_margin[pair];
    GrxMprPairW q;
    for (int k = 0; k < 9; k++) { q.R1[k] = c->gxmat[9 * g1 + k]; q.R2[k] = c->gxmat[9 * g2 + k]; }
    q.t1 = m->geom_type[g1]; q.t2 = m->geom_type[g2]; q.hm = 0.5f * margin; q.lane = lane_;
    for (int k = 0; k < 3; k++) { q.s1[k] = m->geom_size[3 * g1 + k]; q.s2[k] = m->geom_size[3 * g2 + k]; q.c21[k] = (MF)c->gxpos[3 * g2 + k] - (MF)c->gxpos[3 * g1 + k]; }
    q.v1 = q.t1 == 7 ? m->mesh_vert + 3 * m->geom_hulladr[g1] : m->mesh_vert; q.n1 = q.t1 == 7 ? m->geom_hullnum[g1] : 0;
    q.v2 = q.t2 == 7 ? m->mesh_vert + 3 * m->geom_hulladr[g2] : m->mesh_vert; q.n2 = q.t2 == 7 ? m->geom_hullnum[g2] : 0;
    q.aadr1 = m->mesh_adjadr + (q.t1 == 7 ? m->geom_hulladr[g1] : 0); q.anum1 = m->mesh_adjnum + (q.t1 == 7 ? m->geom_hulladr[g1] : 0);
    q.aadr2 = m->mesh_adjadr + (q.t2 == 7 ? m->geom_hulladr[g2] : 0); q.anum2 = m->mesh_adjnum + (q.t2 == 7 ? m->geom_hulladr[g2] : 0); q.adj = m->mesh_adj;
    q.pts = (GrxMprPt*)(c->Jp + 192);
    q.nbr1 = (q.t1 == 7 && m->mesh_nbr) ? m->mesh_nbr + (size_t)4 * GRX_NBR_RECS * m->geom_hulladr[g1] : nullptr;
    q.nbr2 = (q.t2 == 7 && m->mesh_nbr) ? m->mesh_nbr + (size_t)4 * GRX_NBR_RECS * m->geom_hulladr[g2] : nullptr;
    q.hint = 0; q.hk = 0;
    q.cell1 = (q.t1 == 7 && m->mesh_cellhdr && m->geom_cellbase[g1] >= 0) ? m->mesh_cellhdr + 2 * (size_t)m->geom_cellbase[g1] : nullptr;
    q.cell2 = (q.t2 == 7 && m->mesh_cellhdr && m->geom_cellbase[g2] >= 0) ? m->mesh_cellhdr + 2 * (size_t)m->geom_cellbase[g2] : nullptr;
    q.cellrec = m->mesh_cellrec;
#if GRX_DEVICE_PROFILE
    q.prof = c->prof;
#endif   // 30 words behind the pair queue: the Jacobian pool is free until the constraint stage
    GRX_SUBTICK(c, 21);   // pair set-up
    GRX_COUNT(c, 24, 1);
    // a direction kept from an earlier substep: still separating?  (entry: pair + 1, direction, (v1 + 1) + 4096 (v2 + 1) = the support vertices)
    float* mc = c->meshcache;
    const float key = (float)(pair + 1);
    const int slot = mc[0] == key ? 0 : (mc[5] == key ? 1 : (mc[10] == key ? 2 : (mc[15] == key ? 3 : -1)));
    if (slot >= 0) {
      const MF d[3] = {mc[5 * slot + 1], mc[5 * slot + 2], mc[5 * slot + 3]}, nd[3] = {-d[0], -d[1], -d[2]};
      const int hints = (int)mc[5 * slot + 4];
      int h1 = (hints & 4095) - 1, h2 = (hints >> 12) - 1;
      MF sw[3], sb[3], dl[3], r[3];
      if (q.t1 == 7) { mulMatTVec3f(dl, q.R1, d); h1 = grx_mesh_support_hint(m, m->geom_hulladr[g1], q.n1, dl, h1, r, lane_, q.cell1); mulMatVec3f(sw, q.R1, r); }
      else grx_geom_support(q.R1, q.s1, q.t1, d, sw);
      if (q.t2 == 7) { mulMatTVec3f(dl, q.R2, nd); h2 = grx_mesh_support_hint(m, m->geom_hulladr[g2], q.n2, dl, h2, r, lane_, q.cell2); mulMatVec3f(sb, q.R2, r); }
      else grx_geom_support(q.R2, q.s2, q.t2, nd, sb);
      MF sv = 0.0f;   // v . d of the Minkowski support point (see grx_mpr_support)
      for (int k = 0; k < 3; k++) sv += ((sw[k] + d[k] * q.hm) - (sb[k] + q.c21[k] - d[k] * q.hm)) * d[k];
      if (sv < -1e-6f) {   // strictly on the far side: the (inflated) geoms are disjoint
        LANE0 { mc[5 * slot + 4] = (float)((h1 + 1) + 4096 * (h2 + 1)); }
#if !GRX_ON_DEVICE
        g_grx_mesh_stats[0]++;
#endif
        GRX_SUBTICK(c, 22);   // cached separating direction re-checked: disjoint
        continue;
      }
    }
    GRX_SUBTICK(c, 22);
    GRX_COUNT(c, 25, 1);
#if !GRX_ON_DEVICE
    g_grx_mesh_stats[1]++;
#endif
    // guesses of the support vertices, one word per evaluation of this pair's search (the world's HBM row, 4 blocks of key + 16 words): a pair in persistent contact -- the
    // upper arm resting on the head link, the worlds that end a Fetch launch -- repeats its search substep after substep with almost the same directions
    int hblk = -1;
#if GRX_DEVICE_HULL_HINTS
    if (c->hullhint) {
      const float hk0 = c->hullhint[0], hk1 = c->hullhint[17], hk2 = c->hullhint[34], hk3 = c->hullhint[51];
      hblk = hk0 == key ? 0 : (hk1 == key ? 1 : (hk2 == key ? 2 : (hk3 == key ? 3 : -1)));
      hblk = __builtin_amdgcn_readfirstlane(hblk);
      if (hblk >= 0 && lane_ < 16) q.hint = __float_as_int(c->hullhint[17 * hblk + 1 + lane_]);
    }
#elif GRX_TWIN_HULL_HINTS
    for (int l = 0; l < 16; l++) q.hints[l] = 0;
    if (c->hullhint && g_grx_emu_hints_on) {      // emulator twin: the same row, the same four blocks
      const float hk0 = c->hullhint[0], hk1 = c->hullhint[17], hk2 = c->hullhint[34], hk3 = c->hullhint[51];
      hblk = hk0 == key ? 0 : (hk1 == key ? 1 : (hk2 == key ? 2 : (hk3 == key ? 3 : -1)));
      if (hblk >= 0) for (int l = 0; l < 16; l++) memcpy(&q.hints[l], &c->hullhint[17 * hblk + 1 + l], 4);
    }
#endif
    MF depth, dir[3], pos[3], w1[3], w2[3], sep[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const int rc = grx_mpr_penetration<true>(&q, m->mpr_tolerance, m->mpr_iterations, &depth, dir, pos, w1, w2, sep);
#if GRX_DEVICE_HULL_HINTS
    if (c->hullhint && rc == 0) {   // in contact: this search will run again in the next substep
      int wblk = hblk;
      if (wblk < 0) { wblk = ((int)c->hullhint[68]) & 3; if (lane_ == 0) c->hullhint[68] = (float)((wblk + 1) & 3); }
      wblk = __builtin_amdgcn_readfirstlane(wblk);
      if (lane_ < 16) c->hullhint[17 * wblk + 1 + lane_] = __int_as_float(lane_ < q.hk ? q.hint : 0);
      if (lane_ == 0) c->hullhint[17 * wblk] = key;
    }
#elif GRX_TWIN_HULL_HINTS
    if (c->hullhint && g_grx_emu_hints_on && rc == 0) {
      int wblk = hblk;
      if (wblk < 0) { wblk = ((int)c->hullhint[68]) & 3; c->hullhint[68] = (float)((wblk + 1) & 3); }
      for (int l = 0; l < 16; l++) { const int wv = l < q.hk ? q.hints[l] : 0; memcpy(&c->hullhint[17 * wblk + 1 + l], &wv, 4); }
      c->hullhint[17 * wblk] = key;
    }
#endif
    GRX_SUBTICK(c, 23);   // portal search
#ifdef GRX_PROBE_HULL   // outcome of the searches (tools/hull_outcome_probe.py): contacts, separations with a direction, the pair searched last
    GRX_COUNT(c, 35, rc == 0 ? 1 : 0); GRX_COUNT(c, 36, (rc != 0 && sep[3] != 0.0f) ? 1 : 0); GRX_PMAX(c, 37, pair);
#endif
#if GRX_TWIN_MESH_DEBUG
    fprintf(stderr, "meshpair %d (g %d %d) slot %d rc %d sep %g\n", pair, g1, g2, slot, rc, (double)sep[3]);
#endif
    WAVE_SYNC();
    if (rc != 0) {
      if (sep[3] != 0.0f) {   // keep the direction for the next substeps
        const int w = slot >= 0 ? slot : ((int)mc[20] & 3);   // the pair's own slot, else round robin over the four
        LANE0 { mc[5 * w] = key; mc[5 * w + 1] = sep[0]; mc[5 * w + 2] = sep[1]; mc[5 * w + 3] = sep[2]; mc[5 * w + 4] = 0.0f; if (slot < 0) mc[20] = (MF)((w + 1) & 3); }
      }
      WAVE_SYNC();
      continue;
    }
    if (slot >= 0) { LANE0 { mc[5 * slot] = 0.0f; } WAVE_SYNC(); }   // the pair is in contact: its old direction is useless, do not re-check it (two support evaluations) before every search of the next substeps
    if (dir[0] == 0.0f && dir[1] == 0.0f && dir[2] == 0.0f) continue;
    const MF ce1[3] = {0.0f, 0.0f, 0.0f};
    MF n1[3] = {0.0f, 0.0f, 0.0f}, n2[3] = {0.0f, 0.0f, 0.0f};
    const int h1 = grx_smooth_normal(q.R1, ce1, q.s1, q.t1, pos, n1), h2 = grx_smooth_normal(q.R2, q.c21, q.s2, q.t2, pos, n2);
    if (h1 || h2) {   // a smooth primitive against the hull: analytic normal, depth along it (see grx_convex_pair)
      MF n[3] = {n1[0] - n2[0], n1[1] - n2[1], n1[2] - n2[2]};
      const MF l2 = dot3f(n, n);
      if (l2 > 1e-30f) {
        const MF il = 1.0f / grx_sqrt(l2); dir[0] = n[0] * il; dir[1] = n[1] * il; dir[2] = n[2] * il;
        MF nd[3] = {-dir[0], -dir[1], -dir[2]};
        if (h1) { grx_geom_support(q.R1, q.s1, q.t1, dir, w1); for (int k = 0; k < 3; k++) w1[k] += dir[k] * q.hm; }
        if (h2) { grx_geom_support(q.R2, q.s2, q.t2, nd, w2); for (int k = 0; k < 3; k++) w2[k] += q.c21[k] - dir[k] * q.hm; }
        depth = (w1[0] - w2[0]) * dir[0] + (w1[1] - w2[1]) * dir[1] + (w1[2] - w2[2]) * dir[2];
      }
    }
    const float posw[3] = {(float)(pos[0] + c->gxpos[3 * g1]), (float)(pos[1] + c->gxpos[3 * g1 + 1]), (float)(pos[2] + c->gxpos[3 * g1 + 2])}, dirf[3] = {(float)dir[0], (float)dir[1], (float)dir[2]};
    LANE0 { grx_add_contact(c, pair, posw, dirf, (float)(margin - depth)); }
    WAVE_SYNC();
  }
}
// plane vs cylinder: near-cap rim point, far-cap rim point, two more corners of a triangle inscribed in the near rim (see the oracle)
GRX_MEM void grx_plane_cylinder(const GrxModel* m, GrxCtx* c, int pair, int g1, int g2, float margin) {
  const float* pm = c->gxmat + 9 * g1; const float* cm = c->gxmat + 9 * g2; const float* cp = c->gxpos + 3 * g2;
  const float r = m->geom_size[3 * g2], hl = m->geom_size[3 * g2 + 1];
  float n[3] = {pm[2], pm[5], pm[8]}, ax[3] = {cm[2], cm[5], cm[8]};
  float prjaxis = dot3f(n, ax);
  if (prjaxis > 0.0f) { ax[0] = -ax[0]; ax[1] = -ax[1]; ax[2] = -ax[2]; prjaxis = -prjaxis; }
  float dd[3] = {cp[0] - c->gxpos[3 * g1], cp[1] - c->gxpos[3 * g1 + 1], cp[2] - c->gxpos[3 * g1 + 2]};
  const float dist0 = dot3f(dd, n);
  float vec[3] = {ax[0] * prjaxis - n[0], ax[1] * prjaxis - n[1], ax[2] * prjaxis - n[2]};
  const float len2 = dot3f(vec, vec);
  if (len2 >= 1e-30f) { const float sc = r / sqrtf(len2); vec[0] *= sc; vec[1] *= sc; vec[2] *= sc; }
  else { vec[0] = cm[0] * r; vec[1] = cm[3] * r; vec[2] = cm[6] * r; }
  const float prjvec = dot3f(vec, n);
  ax[0] *= hl; ax[1] *= hl; ax[2] *= hl; prjaxis *= hl;
  float dist = dist0 + prjaxis + prjvec, pos[3];
  if (dist > margin) return;
  for (int k = 0; k < 3; k++) pos[k] = cp[k] + vec[k] + ax[k] - n[k] * dist * 0.5f;
  grx_add_contact(c, pair, pos, n, dist);
  dist = dist0 - prjaxis + prjvec;
  if (dist <= margin) {
    for (int k = 0; k < 3; k++) pos[k] = cp[k] + vec[k] - ax[k] - n[k] * dist * 0.5f;
    grx_add_contact(c, pair, pos, n, dist);
  }
  dist = dist0 + prjaxis - 0.5f * prjvec;
  if (dist <= margin) {
    float v1[3];
    cross3f(v1, vec, ax);
    const float l2 = dot3f(v1, v1);
    if (l2 > 0.0f) { const float sc = r * 0.8660254f / sqrtf(l2); v1[0] *= sc; v1[1] *= sc; v1[2] *= sc; }
    for (int sg = 0; sg < 2; sg++) {
      const float sgn = sg ? -1.0f : 1.0f;
      for (int k = 0; k < 3; k++) pos[k] = cp[k] + sgn * v1[k] + ax[k] - 0.5f * vec[k] - n[k] * dist * 0.5f;
      grx_add_contact(c, pair, pos, n, dist);
    }
  }
}
GRX_MEM void grx_plane_ellipsoid(const GrxModel* m, GrxCtx* c, int pair, int g1, int g2, float margin) {
  float n[3] = {c->gxmat[9 * g1 + 2], c->gxmat[9 * g1 + 5], c->gxmat[9 * g1 + 8]}, p[3];
  {
    const MF sz[3] = {m->geom_size[3 * g2], m->geom_size[3 * g2 + 1], m->geom_size[3 * g2 + 2]}, nd[3] = {-n[0], -n[1], -n[2]};
    MF R2[9], pm[3];
    for (int k = 0; k < 9; k++) R2[k] = c->gxmat[9 * g2 + k];
    grx_geom_support(R2, sz, 4, nd, pm);
    p[0] = (float)pm[0]; p[1] = (float)pm[1]; p[2] = (float)pm[2];
  }
  float dd[3];
  for (int k = 0; k < 3; k++) { p[k] += c->gxpos[3 * g2 + k]; dd[k] = p[k] - c->gxpos[3 * g1 + k]; }
  const float dist = dot3f(dd, n);
  if (dist > margin) return;
  float pos[3] = {p[0] - 0.5f * dist * n[0], p[1] - 0.5f * dist * n[1], p[2] - 0.5f * dist * n[2]};
  grx_add_contact(c, pair, pos, n, dist);
}
// capsule (geom1) vs box (geom2): axis point closest to the box (golden-section search, the distance is convex along the
// axis) as a sphere contact, plus the farther end sphere when it is inside the margin as well (see oracle/grx_oracle.c)
GRX_MEM void grx_capsule_box(const GrxModel* m, GrxCtx* c, int pair, int g1, int g2, float margin) {
  const float* ce = c->gxpos + 3 * g1; const float* R = c->gxmat + 9 * g1;
  const float* bp = c->gxpos + 3 * g2; const float* bm = c->gxmat + 9 * g2; const float* sz = m->geom_size + 3 * g2;
  const float r = m->geom_size[3 * g1], hl = m->geom_size[3 * g1 + 1], s0 = sz[0], s1 = sz[1], s2 = sz[2];
  float axw[3] = {R[2], R[5], R[8]}, dw[3] = {ce[0] - bp[0], ce[1] - bp[1], ce[2] - bp[2]}, cen[3], ax[3];
  mulMatTVec3f(cen, bm, dw); mulMatTVec3f(ax, bm, axw);
  // Axis point closest to the box: g(t) = dist^2(box, cen + t ax) is convex and piecewise quadratic, so g'(t)/2 = sum_k ax_k *
  // (p_k - clamp(p_k, -s_k, s_k)) is nondecreasing and piecewise linear with breakpoints where a coordinate crosses a face plane.
  // Evaluate g' at the two ends and the six breakpoints, bracket the sign change between neighbouring candidates, interpolate
  // linearly: the exact minimiser in ~10 evaluations (the oracle finds the same point by golden-section search).
#define GRX_CB_DG(T, OUT) { const float t_ = (T), p0_ = cen[0] + t_ * ax[0], p1_ = cen[1] + t_ * ax[1], p2_ = cen[2] + t_ * ax[2]; \
    OUT = ax[0] * (p0_ - fminf(s0, fmaxf(-s0, p0_))) + ax[1] * (p1_ - fminf(s1, fmaxf(-s1, p1_))) + ax[2] * (p2_ - fminf(s2, fmaxf(-s2, p2_))); }
  float ts, dlo, dhi;
  GRX_CB_DG(-hl, dlo) GRX_CB_DG(hl, dhi)
  if (dlo >= 0.0f) ts = -hl;
  else if (dhi <= 0.0f) ts = hl;
  else {
    float tlo = -hl, thi = hl;   // invariant: g'(tlo) = dlo <= 0 <= dhi = g'(thi)
#define GRX_CB_TRY(TB) { const float tb_ = (TB); if (tb_ > tlo && tb_ < thi) { float d_; GRX_CB_DG(tb_, d_) if (d_ <= 0.0f) { tlo = tb_; dlo = d_; } else { thi = tb_; dhi = d_; } } }
#define GRX_CB_AXIS(K, SK) if (fabsf(ax[K]) > 1e-12f) { const float ia_ = 1.0f / ax[K]; GRX_CB_TRY((SK - cen[K]) * ia_) GRX_CB_TRY((-SK - cen[K]) * ia_) }
    GRX_CB_AXIS(0, s0) GRX_CB_AXIS(1, s1) GRX_CB_AXIS(2, s2)
#undef GRX_CB_AXIS
#undef GRX_CB_TRY
    const float den = dhi - dlo;
    ts = den > 0.0f ? tlo - dlo * (thi - tlo) / den : 0.5f * (tlo + thi);
  }
#undef GRX_CB_DG
  {   // the axis segment passes through the box (penetration deeper than the radius): g vanishes on the whole inside stretch; take its middle
    float ta = -hl, tb = hl; int hit = 1;
#define GRX_CB_SLAB(K, SK) if (fabsf(ax[K]) < GRX_MINVAL) { if (fabsf(cen[K]) > SK) hit = 0; } else { float u_ = (-SK - cen[K]) / ax[K], v_ = (SK - cen[K]) / ax[K]; \
      if (u_ > v_) { const float w_ = u_; u_ = v_; v_ = w_; } ta = fmaxf(ta, u_); tb = fminf(tb, v_); }
    GRX_CB_SLAB(0, s0) GRX_CB_SLAB(1, s1) GRX_CB_SLAB(2, s2)
#undef GRX_CB_SLAB
    if (hit && ta < tb) ts = 0.5f * (ta + tb);
  }
  float ps[3] = {cen[0] + ts * ax[0], cen[1] + ts * ax[1], cen[2] + ts * ax[2]};
  if (!grx_sphere_box_local(c, pair, bp, bm, s0, s1, s2, ps, r, margin)) return;
  float te = (ts >= 0) ? -hl : hl;
  if (fabsf(te - ts) > 0.2f * hl) {
    float pf[3] = {cen[0] + te * ax[0], cen[1] + te * ax[1], cen[2] + te * ax[2]};
    grx_sphere_box_local(c, pair, bp, bm, s0, s1, s2, pf, r, margin);
  }
}

// plane vs a SMALL convex vertex set (<= 32 hull vertices, e.g. the compile-time pruned hulls): one lane does it all
GRX_MEM void grx_plane_mesh_small(const GrxModel* m, GrxCtx* c, int pair, int g1, int g2, float margin) {
  const float* gm = c->gxmat + 9 * g2;
  int adr = m->geom_meshadr[g2], num = m->geom_meshnum[g2];
  float n[3] = {c->gxmat[9 * g1 + 2], c->gxmat[9 * g1 + 5], c->gxmat[9 * g1 + 8]}, nl[3];
  mulMatTVec3f(nl, gm, n);
  float off = dot3f(c->gxpos + 3 * g2, n) - dot3f(c->gxpos + 3 * g1, n);
  float bd = 1e30f; int best = -1;
  // the vertex tables live in global memory: fetch four vertices per round with independent loads (one latency per round, not per vertex)
  for (int v0 = 0; v0 < num; v0 += 4) {
    float vx[4], vy[4], vz[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int v = (v0 + u < num) ? v0 + u : num - 1;
      const float* mv = m->mesh_vert + 3 * (adr + v);
      vx[u] = mv[0]; vy[u] = mv[1]; vz[u] = mv[2];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const float dd = vx[u] * nl[0] + vy[u] * nl[1] + vz[u] * nl[2] + off;
      if (v0 + u < num && dd < bd) { bd = dd; best = v0 + u; }
    }
  }
  if (best < 0 || bd > margin) return;
  // the deepest vertex, then its hull neighbours inside the margin (at most 4 contacts): neighbour indices and their vertices in two rounds
  const int aa = m->mesh_adjadr[adr + best], an = m->mesh_adjnum[adr + best];
  int nb[8]; float wx[8], wy[8], wz[8];
#pragma unroll
  for (int u = 0; u < 8; u++) nb[u] = (u < an) ? m->mesh_adj[aa + u] : best;
#pragma unroll
  for (int u = 0; u < 8; u++) { const float* mv = m->mesh_vert + 3 * (adr + nb[u]); wx[u] = mv[0]; wy[u] = mv[1]; wz[u] = mv[2]; }
  int cn = 0;
  // fully unrolled over the fetched neighbours (static register indices); hull vertices of higher degree take the tail loop
#define GRX_PM_EMIT(LX, LY, LZ, IS_BEST) { \
    const float lx_ = (LX), ly_ = (LY), lz_ = (LZ); \
    const float dd = lx_ * nl[0] + ly_ * nl[1] + lz_ * nl[2] + off; \
    if ((IS_BEST) || dd <= margin) { \
      const float lv[3] = {lx_, ly_, lz_}; float w[3], pos[3]; \
      mulMatVec3f(w, gm, lv); \
      for (int t = 0; t < 3; t++) pos[t] = w[t] + c->gxpos[3 * g2 + t] - 0.5f * dd * n[t]; \
      grx_add_contact(c, pair, pos, n, dd); cn++; \
    } }
  { const float* mv = m->mesh_vert + 3 * (adr + best); GRX_PM_EMIT(mv[0], mv[1], mv[2], 1) }
#define GRX_PM_NB(U) if ((U) < an && cn < 4) GRX_PM_EMIT(wx[U], wy[U], wz[U], 0)
  GRX_PM_NB(0) GRX_PM_NB(1) GRX_PM_NB(2) GRX_PM_NB(3) GRX_PM_NB(4) GRX_PM_NB(5) GRX_PM_NB(6) GRX_PM_NB(7)
#undef GRX_PM_NB
  for (int e = 8; e < an && cn < 4; e++) { const float* mv = m->mesh_vert + 3 * (adr + m->mesh_adj[aa + e]); GRX_PM_EMIT(mv[0], mv[1], mv[2], 0) }
#undef GRX_PM_EMIT
}


// box-box: SAT over the 15 axes, then face clipping or edge-edge (the contact set Sutherland-Hodgman clipping yields:
// (a) incident-face corners inside the reference rectangle, (b) reference corners inside the incident quad, (c) proper
// crossings of incident edges with the rectangle sides; at most 8, in that order).
// Eight lanes work on one pair: lane t evaluates the axes t and t+8, then the contact candidates t, t+8 and t+16; the
// winners are found with DPP reductions inside the octet and the surviving candidates are compacted, in candidate
// order, with wave ballots.  Up to eight pairs per pass; the pair queue is filled by grx_collision.
// Everything stays in registers (no dynamically indexed local arrays): axes are selected with GRX_SEL3.
#define GRX_BB_LOAD(PAIR) /* the pair's record (GrxModel::reci_pair / recf_pair): geoms, margin and both boxes' half sizes in one volley of reads */ \
  const int* PI_ = m->reci_pair + GRX_RPI * (PAIR); const float* PF_ = m->recf_pair + GRX_RPF * (PAIR); \
  const int g1 = PI_[1], g2 = PI_[2]; const float margin = PF_[0]; \
  const float* p1 = c->gxpos + 3 * g1; const float* R1 = c->gxmat + 9 * g1; const float* p2 = c->gxpos + 3 * g2; const float* R2 = c->gxmat + 9 * g2; \
  const float A0[3] = {R1[0], R1[3], R1[6]}, A1[3] = {R1[1], R1[4], R1[7]}, A2[3] = {R1[2], R1[5], R1[8]}; \
  const float B0[3] = {R2[0], R2[3], R2[6]}, B1[3] = {R2[1], R2[4], R2[7]}, B2[3] = {R2[2], R2[5], R2[8]}; \
  const float a0 = PF_[20], a1 = PF_[21], a2 = PF_[22]; \
  const float b0 = PF_[24], b1 = PF_[25], b2 = PF_[26]; \
  const float d[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
// axis T (0-2: faces of box 1, 3-5: faces of box 2, 6-14: edge i of box 1 x edge j of box 2): unit axis, projection of d, separation
#define GRX_BB_AXIS(T, AXV, TP, SEP, OK) { \
  const int T_ = (T), iu_ = T_ < 3 ? T_ : (T_ < 6 ? 0 : (T_ - 6) / 3), jv_ = T_ < 3 ? 0 : (T_ < 6 ? T_ - 3 : (T_ - 6) % 3); \
  float u_[3], v_[3], x_[3]; \
  for (int e_ = 0; e_ < 3; e_++) { u_[e_] = GRX_SEL3(A0[e_], A1[e_], A2[e_], iu_); v_[e_] = GRX_SEL3(B0[e_], B1[e_], B2[e_], jv_); } \
  cross3f(x_, u_, v_); \
  const float l_ = sqrtf(dot3f(x_, x_)), li_ = 1.0f / fmaxf(l_, 1e-12f); \
  OK = (T_ < 15) && ((T_ < 6) || (l_ >= 1e-6f)); \
  for (int e_ = 0; e_ < 3; e_++) AXV[e_] = T_ < 3 ? u_[e_] : (T_ < 6 ? v_[e_] : x_[e_] * li_); \
  TP = dot3f(d, AXV); \
  float ra_ = a0 * fabsf(dot3f(A0, AXV)) + a1 * fabsf(dot3f(A1, AXV)) + a2 * fabsf(dot3f(A2, AXV)); \
  float rb_ = b0 * fabsf(dot3f(B0, AXV)) + b1 * fabsf(dot3f(B1, AXV)) + b2 * fabsf(dot3f(B2, AXV)); \
  if (T_ < 3) ra_ = GRX_SEL3(a0, a1, a2, T_); else if (T_ < 6) rb_ = GRX_SEL3(b0, b1, b2, T_ - 3); \
  SEP = fabsf(TP) - (ra_ + rb_); }

GRX_MEM void grx_box_box_queue(const GrxModel* m, GrxCtx* c, const int* queue, int nq, int lane_) {
  for (int pb = 0; pb < nq; pb += 8) {
    // ---- separating axes
    GRX_LANEVAR(sf); GRX_LANEVAR(se); GRX_LANEVAR(sall); GRX_LANEVAR_I(cf); GRX_LANEVAR_I(ce);
    FOR_LANES {
      const int g = lane >> 3, t = lane & 7;
      float f = -1e30f, e = -1e30f; int fi = 99, ei = 99;
      if (pb + g < nq) {
        const int pair = queue[pb + g];
        GRX_BB_LOAD(pair)
        float ax[3], tp, sep; int ok;
        GRX_BB_AXIS(t, ax, tp, sep, ok)
        if (ok) { if (t < 6) { f = sep; fi = t; } else { e = sep; ei = t; } }
        GRX_BB_AXIS(t + 8, ax, tp, sep, ok)
        if (ok && sep > e) { e = sep; ei = t + 8; }
        (void)tp; (void)margin;
      }
      LV(sf) = f; LV(se) = e; LV(cf) = fi; LV(ce) = ei; LV(sall) = fmaxf(f, e);
    }
    GRX_LANEVAR(bestf); GRX_LANEVAR(beste); GRX_LANEVAR(maxall); GRX_LANEVAR_I(codef); GRX_LANEVAR_I(codee);
    GRX_OCT_MAX(sf, bestf); GRX_OCT_MAX(se, beste); GRX_OCT_MAX(sall, maxall);
    FOR_LANES { if (!(LV(sf) == LV(bestf))) LV(cf) = 99; if (!(LV(se) == LV(beste))) LV(ce) = 99; }
    GRX_OCT_MIN_I(cf, codef); GRX_OCT_MIN_I(ce, codee);
    // ---- contact candidates
    GRX_LANEVAR(nx); GRX_LANEVAR(ny); GRX_LANEVAR(nz);
    GRX_LANEVAR(cpx0); GRX_LANEVAR(cpy0); GRX_LANEVAR(cpz0); GRX_LANEVAR(ch0); GRX_LANEVAR_I(cv0);
    GRX_LANEVAR(cpx1); GRX_LANEVAR(cpy1); GRX_LANEVAR(cpz1); GRX_LANEVAR(ch1); GRX_LANEVAR_I(cv1);
    GRX_LANEVAR(cpx2); GRX_LANEVAR(cpy2); GRX_LANEVAR(cpz2); GRX_LANEVAR(ch2); GRX_LANEVAR_I(cv2);
    FOR_LANES {
      const int g = lane >> 3, t = lane & 7;
      int v0 = 0, v1 = 0, v2 = 0; float P0[3] = {0, 0, 0}, P1[3] = {0, 0, 0}, P2[3] = {0, 0, 0}, h0 = 0, h1 = 0, h2 = 0, nrm[3] = {0, 0, 0};
      if (pb + g < nq) {
        const int pair = queue[pb + g];
        GRX_BB_LOAD(pair)
        const float best = LV(bestf), ebest = LV(beste); const int code = LV(codef), ecode = LV(codee);
        if (LV(maxall) <= margin && code < 6) {
          if (ecode < 15 && ebest > best + 1e-7f + 0.02f * fabsf(best)) {
            // edge-edge: a single contact, lane 0 of the octet
            const int ei = (ecode - 6) / 3, ej = (ecode - 6) % 3;
            float en[3], tp, sep; int ok;
            GRX_BB_AXIS(ecode, en, tp, sep, ok)
            (void)sep; (void)ok;
            const float sg = tp < 0 ? -1.0f : 1.0f;
            en[0] *= sg; en[1] *= sg; en[2] *= sg;
            float pa[3] = {p1[0], p1[1], p1[2]}, pb_[3] = {p2[0], p2[1], p2[2]};
            float s0 = (ei != 0) ? (dot3f(en, A0) > 0 ? a0 : -a0) : 0.0f, s1 = (ei != 1) ? (dot3f(en, A1) > 0 ? a1 : -a1) : 0.0f, s2 = (ei != 2) ? (dot3f(en, A2) > 0 ? a2 : -a2) : 0.0f;
            float t0 = (ej != 0) ? (dot3f(en, B0) > 0 ? -b0 : b0) : 0.0f, t1 = (ej != 1) ? (dot3f(en, B1) > 0 ? -b1 : b1) : 0.0f, t2 = (ej != 2) ? (dot3f(en, B2) > 0 ? -b2 : b2) : 0.0f;
            float u[3], v[3];
            for (int e = 0; e < 3; e++) {
              pa[e] += s0 * A0[e] + s1 * A1[e] + s2 * A2[e]; pb_[e] += t0 * B0[e] + t1 * B1[e] + t2 * B2[e];
              u[e] = GRX_SEL3(A0[e], A1[e], A2[e], ei); v[e] = GRX_SEL3(B0[e], B1[e], B2[e], ej);
            }
            float w[3] = {pa[0] - pb_[0], pa[1] - pb_[1], pa[2] - pb_[2]};
            float uv = dot3f(u, v), uw = dot3f(u, w), vw = dot3f(v, w);
            float den = 1.0f - uv * uv;
            float sc = den > 1e-12f ? (uv * vw - uw) / den : 0.0f, tc = den > 1e-12f ? (vw - uv * uw) / den : 0.0f;
            for (int k = 0; k < 3; k++) { P0[k] = 0.5f * ((pa[k] + sc * u[k]) + (pb_[k] + tc * v[k])); nrm[k] = en[k]; }
            h0 = ebest; v0 = (t == 0);
          } else {
            // ---- face contact
            float bn[3], tp, sep; int ok;
            GRX_BB_AXIS(code, bn, tp, sep, ok)
            (void)sep; (void)ok;
            { const float sg = tp < 0 ? -1.0f : 1.0f; bn[0] *= sg; bn[1] *= sg; bn[2] *= sg; }
            const int ref1 = code < 3, ax = ref1 ? code : code - 3;
            float pr[3], pi[3], nr[3], Ar0[3], Ar1[3], Ar2[3], Ai0[3], Ai1[3], Ai2[3];
            for (int e = 0; e < 3; e++) {
              pr[e] = ref1 ? p1[e] : p2[e]; pi[e] = ref1 ? p2[e] : p1[e]; nr[e] = ref1 ? bn[e] : -bn[e];
              Ar0[e] = ref1 ? A0[e] : B0[e]; Ar1[e] = ref1 ? A1[e] : B1[e]; Ar2[e] = ref1 ? A2[e] : B2[e];
              Ai0[e] = ref1 ? B0[e] : A0[e]; Ai1[e] = ref1 ? B1[e] : A1[e]; Ai2[e] = ref1 ? B2[e] : A2[e];
            }
            const float sr0 = ref1 ? a0 : b0, sr1 = ref1 ? a1 : b1, sr2 = ref1 ? a2 : b2;
            const float si0 = ref1 ? b0 : a0, si1 = ref1 ? b1 : a1, si2 = ref1 ? b2 : a2;
            // incident face: the face of the other box most anti-parallel to nr
            float dd0 = dot3f(Ai0, nr), dd1 = dot3f(Ai1, nr), dd2 = dot3f(Ai2, nr);
            int iax = 0; float mind = 1e30f, isg = 1.0f;
            if (dd0 < mind) { mind = dd0; iax = 0; isg = 1.0f; } if (-dd0 < mind) { mind = -dd0; iax = 0; isg = -1.0f; }
            if (dd1 < mind) { mind = dd1; iax = 1; isg = 1.0f; } if (-dd1 < mind) { mind = -dd1; iax = 1; isg = -1.0f; }
            if (dd2 < mind) { mind = dd2; iax = 2; isg = 1.0f; } if (-dd2 < mind) { mind = -dd2; iax = 2; isg = -1.0f; }
            float Iu[3], Iv[3], In[3], Ru[3], Rv[3];
            for (int e = 0; e < 3; e++) {
              In[e] = GRX_SEL3(Ai0[e], Ai1[e], Ai2[e], iax); Iu[e] = GRX_SEL3(Ai1[e], Ai2[e], Ai0[e], iax); Iv[e] = GRX_SEL3(Ai2[e], Ai0[e], Ai1[e], iax);
              Ru[e] = GRX_SEL3(Ar1[e], Ar2[e], Ar0[e], ax); Rv[e] = GRX_SEL3(Ar2[e], Ar0[e], Ar1[e], ax);
            }
            const float sin_ = GRX_SEL3(si0, si1, si2, iax), siu = GRX_SEL3(si1, si2, si0, iax), siv = GRX_SEL3(si2, si0, si1, iax);
            const float srn = GRX_SEL3(sr0, sr1, sr2, ax), sx = GRX_SEL3(sr1, sr2, sr0, ax), sy = GRX_SEL3(sr2, sr0, sr1, ax);
            float rc[3], fcw[3];
            for (int e = 0; e < 3; e++) { rc[e] = pr[e] + srn * nr[e]; fcw[e] = pi[e] + isg * sin_ * In[e] - rc[e]; }
            // incident quad in the reference face frame: corner q = centre + su*U + sv*V, (su,sv) = (+,+),(-,+),(-,-),(+,-)
            const float cx = dot3f(fcw, Ru), cy = dot3f(fcw, Rv), chh = dot3f(fcw, nr);
            const float ux = siu * dot3f(Iu, Ru), uy = siu * dot3f(Iu, Rv), uh = siu * dot3f(Iu, nr);
            const float vx = siv * dot3f(Iv, Ru), vy = siv * dot3f(Iv, Rv), vh = siv * dot3f(Iv, nr);
            const float qx0 = cx + ux + vx, qy0 = cy + uy + vy, qh0 = chh + uh + vh;
            const float qx1 = cx - ux + vx, qy1 = cy - uy + vy, qh1 = chh - uh + vh;
            const float qx2 = cx - ux - vx, qy2 = cy - uy - vy;
            const float qx3 = cx + ux - vx, qy3 = cy + uy - vy, qh3 = chh + uh - vh;
            // height field of the incident plane over the reference frame
            const float x1 = qx1 - qx0, y1 = qy1 - qy0, x2 = qx3 - qx0, y2 = qy3 - qy0, hh1 = qh1 - qh0, hh2 = qh3 - qh0;
            const float det = x1 * y2 - x2 * y1;
            const int flat = !(fabsf(det) > 1e-14f);
            const float gu = flat ? 0.0f : (hh1 * y2 - hh2 * y1) / det, gv = flat ? 0.0f : (x1 * hh2 - x2 * hh1) / det;
            const float orient = det > 0 ? -1.0f : 1.0f;  // det > 0 <=> q0->q1->q2->q3 is counter-clockwise <=> interior has cross > 0
#define GRX_SEL4(v0_, v1_, v2_, v3_, i_) ((i_) == 0 ? (v0_) : ((i_) == 1 ? (v1_) : ((i_) == 2 ? (v2_) : (v3_))))
#define GRX_SIDE(PX, PY, AX_, AY_, BX_, BY_) (orient * (((BX_) - (AX_)) * ((PY) - (AY_)) - ((BY_) - (AY_)) * ((PX) - (AX_))))
            // candidate I: 0-3 incident corners inside the rectangle (inclusive); 4-7 rectangle corners strictly inside the
            // incident quad; 8-23 proper crossings of incident edge e = (I-8)/4 with rectangle side (I-8)%4 = +x, -x, +y, -y
            // (x-sides closed in y, y-sides open in x)
#define GRX_BB_CAND(I, VALID, POS, H) { \
              const int i_ = (I); int ok_ = 0; float X_ = 0, Y_ = 0; \
              if (i_ < 4) { X_ = GRX_SEL4(qx0, qx1, qx2, qx3, i_); Y_ = GRX_SEL4(qy0, qy1, qy2, qy3, i_); ok_ = fabsf(X_) <= sx && fabsf(Y_) <= sy; } \
              else if (!flat && i_ < 8) { \
                const int k_ = i_ - 4; X_ = (k_ == 0 || k_ == 3) ? sx : -sx; Y_ = (k_ < 2) ? sy : -sy; \
                ok_ = GRX_SIDE(X_, Y_, qx0, qy0, qx1, qy1) < 0 && GRX_SIDE(X_, Y_, qx1, qy1, qx2, qy2) < 0 && GRX_SIDE(X_, Y_, qx2, qy2, qx3, qy3) < 0 && \
                      GRX_SIDE(X_, Y_, qx3, qy3, qx0, qy0) < 0; \
              } else if (!flat && i_ < 24) { \
                const int e_ = (i_ - 8) >> 2, s_ = (i_ - 8) & 3, e1_ = (e_ + 1) & 3; \
                const float ax_ = GRX_SEL4(qx0, qx1, qx2, qx3, e_), ay_ = GRX_SEL4(qy0, qy1, qy2, qy3, e_); \
                const float bx_ = GRX_SEL4(qx0, qx1, qx2, qx3, e1_), by_ = GRX_SEL4(qy0, qy1, qy2, qy3, e1_); \
                const int hz_ = s_ < 2; const float sg_ = (s_ & 1) ? -1.0f : 1.0f; \
                const float pa_ = hz_ ? ax_ : ay_, pb2_ = hz_ ? bx_ : by_, lim_ = hz_ ? sx : sy; \
                const float da_ = sg_ * pa_ - lim_, db_ = sg_ * pb2_ - lim_; \
                const int cr_ = (da_ < 0 && db_ > 0) || (da_ > 0 && db_ < 0); \
                const float t_ = da_ / (da_ - db_), oa_ = hz_ ? ay_ : ax_, ob_ = hz_ ? by_ : bx_, o_ = oa_ + t_ * (ob_ - oa_); \
                if (hz_) { ok_ = cr_ && fabsf(o_) <= sy; X_ = sg_ * sx; Y_ = o_; } else { ok_ = cr_ && fabsf(o_) < sx; X_ = o_; Y_ = sg_ * sy; } \
              } \
              const float h_ = qh0 + gu * (X_ - qx0) + gv * (Y_ - qy0); \
              VALID = ok_ && h_ <= margin; H = h_; \
              for (int k_ = 0; k_ < 3; k_++) POS[k_] = rc[k_] + X_ * Ru[k_] + Y_ * Rv[k_] + 0.5f * h_ * nr[k_]; }
            GRX_BB_CAND(t, v0, P0, h0)
            GRX_BB_CAND(t + 8, v1, P1, h1)
            GRX_BB_CAND(t + 16, v2, P2, h2)
#undef GRX_BB_CAND
#undef GRX_SIDE
#undef GRX_SEL4
            for (int k = 0; k < 3; k++) nrm[k] = bn[k];
          }
        }
      }
      LV(nx) = nrm[0]; LV(ny) = nrm[1]; LV(nz) = nrm[2];
      LV(cpx0) = P0[0]; LV(cpy0) = P0[1]; LV(cpz0) = P0[2]; LV(ch0) = h0; LV(cv0) = v0;
      LV(cpx1) = P1[0]; LV(cpy1) = P1[1]; LV(cpz1) = P1[2]; LV(ch1) = h1; LV(cv1) = v1;
      LV(cpx2) = P2[0]; LV(cpy2) = P2[1]; LV(cpz2) = P2[2]; LV(ch2) = h2; LV(cv2) = v2;
    }
    // ---- ordered compaction: candidate order inside a pair, pair order across the octets, at most 8 contacts per pair
    const unsigned long long m0 = GRX_BALLOT(cv0), m1 = GRX_BALLOT(cv1), m2 = GRX_BALLOT(cv2);
    WAVE_SYNC();
    const int base = c->cnt[0];
    int total = 0;
    for (int g = 0; g < 8; g++) { int n = __builtin_popcountll((m0 >> (8 * g)) & 0xFFull) + __builtin_popcountll((m1 >> (8 * g)) & 0xFFull) + __builtin_popcountll((m2 >> (8 * g)) & 0xFFull); total += n < 8 ? n : 8; }
    FOR_LANES {
      const int g = lane >> 3, t = lane & 7;
      if (pb + g < nq) {
        const int pair = queue[pb + g];
        int gbase = base;
        for (int q = 0; q < g; q++) { int n = __builtin_popcountll((m0 >> (8 * q)) & 0xFFull) + __builtin_popcountll((m1 >> (8 * q)) & 0xFFull) + __builtin_popcountll((m2 >> (8 * q)) & 0xFFull); gbase += n < 8 ? n : 8; }
        const unsigned b0_ = (unsigned)((m0 >> (8 * g)) & 0xFFull), b1_ = (unsigned)((m1 >> (8 * g)) & 0xFFull), b2_ = (unsigned)((m2 >> (8 * g)) & 0xFFull), low = (1u << t) - 1u;
        const int r0 = __builtin_popcount(b0_ & low), r1 = __builtin_popcount(b0_) + __builtin_popcount(b1_ & low), r2 = __builtin_popcount(b0_) + __builtin_popcount(b1_) + __builtin_popcount(b2_ & low);
        const float nrm[3] = {LV(nx), LV(ny), LV(nz)};
#define GRX_BB_WRITE(V, R, PX, PY, PZ, H) if ((V) && (R) < 8) { const int slot = gbase + (R); \
          if (slot >= c->maxcon) c->cnt[2] |= GRX_ST_CON_OVERFLOW; \
          else { c->con_dist[slot] = (H); c->con_pair[slot] = pair; c->con_pos[3 * slot] = (PX); c->con_pos[3 * slot + 1] = (PY); c->con_pos[3 * slot + 2] = (PZ); \
                 for (int k_ = 0; k_ < 3; k_++) c->con_frame[3 * slot + k_] = nrm[k_]; } }
        GRX_BB_WRITE(LV(cv0), r0, LV(cpx0), LV(cpy0), LV(cpz0), LV(ch0))
        GRX_BB_WRITE(LV(cv1), r1, LV(cpx1), LV(cpy1), LV(cpz1), LV(ch1))
        GRX_BB_WRITE(LV(cv2), r2, LV(cpx2), LV(cpy2), LV(cpz2), LV(ch2))
#undef GRX_BB_WRITE
      }
    }
    WAVE_SYNC();
    LANE0 { c->cnt[0] = base + total; }
    WAVE_SYNC();
  }
}
#undef GRX_BB_AXIS
#undef GRX_BB_LOAD

GRX_MEM void grx_collision(const GrxModel* m, GrxCtx* c, int lane_) {
  GRX_OPAQUE_STAGE(lane_);
  GRX_FRESH_MODEL(m, c);
  // geom frames (they share LDS with the composite inertias of the previous stage)
  FOR_LANES {
    for (int i = lane; i < GRX_NGC; i += 64) {
      int b = m->geom_bodyid[i];
      float lpv[3] = {m->geom_pos[3 * i], m->geom_pos[3 * i + 1], m->geom_pos[3 * i + 2]}, lqv[4] = {m->geom_quat[4 * i], m->geom_quat[4 * i + 1], m->geom_quat[4 * i + 2], m->geom_quat[4 * i + 3]}, v[3], R[9], Rw[9];
      mulMatVec3f(v, c->xmat + 9 * b, lpv);
      const int sh = (S::kShift && m->nshift) ? m->geom_shift[i] : 0;
      for (int e = 0; e < 3; e++) v[e] += c->xpos[3 * b + e];
      quat2matf(R, lqv); mulMat3f(Rw, c->xmat + 9 * b, R);
      if (S::kShiftRot && sh == 2) grx_apply_group_rotation(c->shift + 3, v, Rw);
      for (int e = 0; e < 3; e++) c->gxpos[3 * i + e] = v[e] + (sh ? c->shift[e] : 0.0f);
      for (int e = 0; e < 9; e++) c->gxmat[9 * i + e] = Rw[e];
    }
  }
  LANE0 { c->cnt[0] = 0; c->cnt[7] = 0; }
  WAVE_SYNC();
  GRX_RNDINJ(7, (grx_rnd(c->gxpos, 3 * m->ngeom), grx_rnd(c->gxmat, 9 * m->ngeom)));
  GRX_SUBTICK(c, 12);
  // Wall lattice (maze layouts): a moving sphere / capsule only meets the walls of the 3 x 3 cells around its centre -- nine table lookups per
  // mover instead of one bounding-sphere test per (mover, wall) pair of the flat list; same pairs, same tests, same narrow phase.
  if (m->ngridgeom > 0) {
    FOR_LANES {
      for (int it = lane; it < 9 * m->ngridgeom; it += 64) {
        const int k = it / 9, nb = it - 9 * k, rec = m->grid_geom[k], g1 = rec & 0xFFF, t1 = rec >> 12;
        const float r = m->grid_geom_bound[2 * k], margin = m->grid_geom_bound[2 * k + 1];
        const int ix = (int)floorf((c->gxpos[3 * g1] - m->gridx0) * m->gridinv) + (nb % 3) - 1, iy = (int)floorf((c->gxpos[3 * g1 + 1] - m->gridy0) * m->gridinv) + (nb / 3) - 1;
        if (ix >= 0 && iy >= 0 && ix < m->gridnx && iy < m->gridny) {
          const int wl = m->grid_cell[iy * m->gridnx + ix];
          if (wl >= 0) {
            const int g2 = m->grid_wall_geom[wl], p = m->grid_pair[k * m->ngridwall + wl];
            float dx[3] = {c->gxpos[3 * g2] - c->gxpos[3 * g1], c->gxpos[3 * g2 + 1] - c->gxpos[3 * g1 + 1], c->gxpos[3 * g2 + 2] - c->gxpos[3 * g1 + 2]};
            if (dot3f(dx, dx) <= r * r) {
              if (t1 == 2) grx_sphere_box(m, c, p, g1, g2, margin);
              else grx_capsule_box(m, c, p, g1, g2, margin);
            }
          }
        }
      }
    }
    WAVE_SYNC();
  }
  // Models with more than one wave of candidates (the Fetch arm: 163, most of them hull pairs): a first sweep runs only the bounding-sphere /
  // plane-distance test and compacts the survivors (ballot prefix, pair order), so the narrow phases and the bounding-box tests below see
  // ONE dense pass instead of one sparse, divergent pass per 64 candidates.
  // Scenes with more candidates than the survivor list has room for (the kitchen: 3 736) are swept in chunks of that size: pair order is kept.
  const int ndp = m->ndevpair;
  const bool kGate = S::kHullFilter && m->ngate > 0;   // joint-box gates of hull pairs (grx_gate_clear); the skin-list sweep of the large scenes does not use them
  unsigned long long gmask = 0ull;   // the model's gates (at most 64: the compiler keeps those of the nearest pairs) evaluated once per pass, one lane each; the sweep tests a bit
  if (kGate) { GRX_LANEVAR_I(gc); FOR_LANES { LV(gc) = (lane < m->ngate) ? grx_gate_clear(m, c, lane) : 0; } gmask = GRX_BALLOT(gc); }
#define GRX_GATE_CLEAR(gi) ((int)((gmask >> ((gi) & 63)) & 1ull))
  constexpr bool kChunked = !S::kFixed || S::NG > 64;   // small scenes (every specialised shape but the kitchen): one pass, no loop around the sweep
  // Skin list (large scenes, GPU build): the kitchen has 3 736 candidate pairs of which ~170 pass the bounding-sphere test and ~250 are within 10 cm of
  // passing it.  The flat sweep of all candidates in every substep is replaced by a sweep of the pairs that passed the test with the radius inflated
  // by `skin` when the list was built; the list (pair order) and the geom positions at that moment live in HBM, one row per world, across substeps
  // AND launches.  Every substep checks the largest geom displacement since the build: while 2 * displacement <= skin no pair outside the list can pass
  // the exact test (which only reads the two geom centres; plane geoms are static, checked by the host), so the survivors -- and everything after
  // them -- are exactly those of the full sweep.  Otherwise (and for a zeroed row) the list is rebuilt first: one full sweep per ~40 substeps.
  const int* slist = nullptr; int ncand = ndp;
#if GRX_ON_DEVICE
  if (kChunked && c->skin != nullptr && ndp > 256) {
    volatile int* hdr = c->skin; float* gref = (float*)(c->skin + 4); int* list = c->skin + 4 + 3 * GRX_NGC;
    const float skin = c->skin_r;
    float d2 = 0.0f;
    for (int g = lane_; g < GRX_NGC; g += 64) {
      const float dx = c->gxpos[3 * g] - gref[3 * g], dy = c->gxpos[3 * g + 1] - gref[3 * g + 1], dz = c->gxpos[3 * g + 2] - gref[3 * g + 2];
      d2 = fmaxf(d2, dx * dx + dy * dy + dz * dz);
    }
    const int valid = hdr[1];
    const float dmax2 = grx_reduce_max(d2);
    if (!valid || !(4.0f * dmax2 <= 0.81f * skin * skin)) {   // 10 % of the skin left for the rounding of the two tests
      for (int g = lane_; g < 3 * GRX_NGC; g += 64) gref[g] = c->gxpos[g];
      int ns = 0;
      for (int base = 0; base < ndp; base += 256) {
        unsigned rec[4]; float mg[4], rb[4]; int pass[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int k = base + 64 * u + lane_, kk = k < ndp ? k : 0; rec[u] = (unsigned)m->devpair_geoms[kk]; mg[u] = m->devpair_bound[2 * kk]; rb[u] = m->devpair_bound[2 * kk + 1]; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int g1 = rec[u] & 0xFFF, g2 = (rec[u] >> 12) & 0xFFF, t1 = (rec[u] >> 24) & 0xF;
          float dx[3] = {c->gxpos[3 * g2] - c->gxpos[3 * g1], c->gxpos[3 * g2 + 1] - c->gxpos[3 * g1 + 1], c->gxpos[3 * g2 + 2] - c->gxpos[3 * g1 + 2]};
          const float r = rb[u] + mg[u] + skin;
          int ps;
          if (t1 == 0) { float n[3] = {c->gxmat[9 * g1 + 2], c->gxmat[9 * g1 + 5], c->gxmat[9 * g1 + 8]}; ps = dot3f(dx, n) <= r; }
          else ps = dot3f(dx, dx) <= r * r;
          pass[u] = (base + 64 * u + lane_ < ndp) && ps;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const unsigned long long bm = __ballot(pass[u]);
          if (pass[u]) list[ns + __builtin_popcountll(bm & ((1ull << lane_) - 1ull))] = base + 64 * u + lane_;
          ns += __builtin_popcountll(bm);
        }
      }
      if (lane_ == 0) { hdr[0] = ns; hdr[1] = 1; }
      __threadfence_block();
      WAVE_SYNC();
    }
    ncand = hdr[0]; slist = list;
  }
#endif
  GRX_SUBTICK(c, 36);        // profiling build: skin-list validity check (+ the rebuilds) apart from the sweep of the listed pairs
  GRX_COUNT(c, 37, ncand);   // ... candidates swept per substep (summed over the step)
  const int cap = c->jpool - 256, compact = ncand > 64 && cap >= 64;
  const bool classed = kChunked && ndp > 512 && ndp < 65536;   // a property of the MODEL (not of the kernel shape): the generic and the specialised kernel agree
  const int chunk = (kChunked && compact && cap < ncand) ? cap : (ncand > 0 ? ncand : 1);
  int c0 = 0;
  do {
  const int cend = c0 + chunk < ncand ? c0 + chunk : ncand;
  int nsurv = cend - c0; const int* surv = nullptr;
  if (compact) {
    int* sv = (int*)(c->Jp + 256);   // the Jacobian pool is free until the constraint stage ([0, 128) is c->red, [128, 222) the hull-pair queue + portal)
    int ns = 0;
    // four groups of 64 candidates per round: their table records (global memory, one dependent load chain per candidate) are fetched together,
    // so that a round pays one memory latency instead of four
    for (int base = c0; base < cend; base += 256) {
      GRX_LANEVAR_I(ps0); GRX_LANEVAR_I(ps1); GRX_LANEVAR_I(ps2); GRX_LANEVAR_I(ps3);
      GRX_LANEVAR_I(kp0); GRX_LANEVAR_I(kp1); GRX_LANEVAR_I(kp2); GRX_LANEVAR_I(kp3);
      FOR_LANES {
        unsigned rec[4]; float mg[4], rb[4]; int ok[4], kp[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int k = base + 64 * u + lane;
          ok[u] = k < cend;
          kp[u] = ok[u] ? k : c0;
          if (kChunked && slist) kp[u] = slist[kp[u]];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) { const int kk = kp[u]; rec[u] = (unsigned)m->devpair_geoms[kk]; mg[u] = m->devpair_bound[2 * kk]; rb[u] = m->devpair_bound[2 * kk + 1]; }
        int gate[4];
#pragma unroll
        for (int u = 0; u < 4; u++) gate[u] = kGate ? m->devpair_gate[kp[u]] : -1;
        int pass[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int g1 = rec[u] & 0xFFF, g2 = (rec[u] >> 12) & 0xFFF, t1 = (rec[u] >> 24) & 0xF;
          float dx[3] = {c->gxpos[3 * g2] - c->gxpos[3 * g1], c->gxpos[3 * g2 + 1] - c->gxpos[3 * g1 + 1], c->gxpos[3 * g2 + 2] - c->gxpos[3 * g1 + 2]};
          int ps;
          if (t1 == 0) { float n[3] = {c->gxmat[9 * g1 + 2], c->gxmat[9 * g1 + 5], c->gxmat[9 * g1 + 8]}; ps = dot3f(dx, n) <= rb[u] + mg[u]; }
          else { const float r = rb[u] + mg[u]; ps = dot3f(dx, dx) <= r * r; }
          pass[u] = ok[u] && ps;
          if (kGate && pass[u] && gate[u] >= 0 && GRX_GATE_CLEAR(gate[u])) pass[u] = 0;   // proven disjoint at these joint values
          if (classed) {
            const int t2 = rec[u] >> 28;
            // Large scenes, second filter: the bounding spheres of long thin geoms are loose (the kitchen: ~120 capsule-box candidates per substep pass
            // the sphere test, none of them passes this one) -- separating-axis test of the two oriented bounding boxes, grown by the margin plus a
            // rounding allowance: a pair it rejects cannot produce a contact within the margin, so the contact list does not change.
            if (pass[u] && t1 >= 3 && t2 >= 3) pass[u] = grx_obb_overlap(m, c, g1, g2, mg[u] + 4e-6f);
            // kind of narrow phase (see the regrouping below), carried in the bits above the pair index
            kp[u] |= ((t2 == 7 && t1 != 0) ? 2 : ((t1 == 6 && t2 == 6) ? 1 : ((S::kConvex && t1 >= 2 && t2 <= 6 && (t1 == 4 || t1 == 5 || t2 == 4 || t2 == 5)) ? 3 : 0))) << 16;
          }
        }
        LV(ps0) = pass[0]; LV(ps1) = pass[1]; LV(ps2) = pass[2]; LV(ps3) = pass[3];
        LV(kp0) = kp[0]; LV(kp1) = kp[1]; LV(kp2) = kp[2]; LV(kp3) = kp[3];
      }
#define GRX_COMPACT_GROUP(PS, KP) { const unsigned long long bm = GRX_BALLOT(PS); \
        FOR_LANES { if (LV(PS)) sv[ns + __builtin_popcountll(bm & ((1ull << lane) - 1ull))] = LV(KP); } \
        ns += __builtin_popcountll(bm); }
      GRX_COMPACT_GROUP(ps0, kp0) GRX_COMPACT_GROUP(ps1, kp1) GRX_COMPACT_GROUP(ps2, kp2) GRX_COMPACT_GROUP(ps3, kp3)
#undef GRX_COMPACT_GROUP
    }
    WAVE_SYNC();
    nsurv = ns; surv = sv;
    GRX_SUBTICK(c, 19);
    GRX_COUNT(c, 38, ns);    // profiling build: survivors of the sweep (sphere / plane test + bounding-box filter)
    // Large scenes: the survivors (the kitchen: ~170 per substep, three rounds of 64) are regrouped by the KIND of narrow phase they need -- analytic
    // primitive tests, box-box (queued), hull pairs (bounding-box test + queue), portal refinement on a lane -- so that a round of 64 lanes runs one
    // kind instead of paying every kind's divergent branch in every round.  The contact list is put back into pair order afterwards (below).
    if (classed && ns > 64 && 2 * ns <= cap) {
      int* dst = sv + ns;
      int off[4] = {0, 0, 0, 0};
      for (int base = 0; base < ns; base += 64) {
        GRX_LANEVAR_I(cl);
        FOR_LANES { LV(cl) = base + lane < ns ? (sv[base + lane] >> 16) : -1; }
        for (int q = 0; q < 3; q++) { GRX_LANEVAR_I(hit); FOR_LANES { LV(hit) = LV(cl) == q; } off[q + 1] += __builtin_popcountll(GRX_BALLOT(hit)); }
      }
      off[3] += off[2] + off[1]; off[2] += off[1];     // counts of the classes 0 .. 2 -> start of the classes 1 .. 3
      for (int base = 0; base < ns; base += 64) {
        GRX_LANEVAR_I(cl);
        FOR_LANES { LV(cl) = base + lane < ns ? (sv[base + lane] >> 16) : -1; }
        for (int q = 0; q < 4; q++) {
          GRX_LANEVAR_I(hit);
          FOR_LANES { LV(hit) = LV(cl) == q; }
          const unsigned long long bm = GRX_BALLOT(hit);
          FOR_LANES { if (LV(hit)) dst[off[q] + __builtin_popcountll(bm & ((1ull << lane) - 1ull))] = sv[base + lane]; }
          off[q] += __builtin_popcountll(bm);
        }
      }
      WAVE_SYNC();
      surv = dst;
    }
    GRX_SUBTICK(c, 20);
  }
  for (int base = 0; base < nsurv; base += 64) {
    GRX_LANEVAR_I(boxq); GRX_LANEVAR_I(meshq); GRX_LANEVAR_I(pairq);
    FOR_LANES {
      int isbox = 0, ismesh = 0, pq = 0;
      if (base + lane < nsurv) {
        const int k = surv ? (surv[base + lane] & 0xFFFF) : ((kChunked && slist) ? slist[base + lane] : base + lane);
        // one packed record per candidate (geoms, types, margin, broad-phase radius): a single level of model-table loads
        const unsigned rec = (unsigned)m->devpair_geoms[k];
        const int g1 = rec & 0xFFF, g2 = (rec >> 12) & 0xFFF, t1 = (rec >> 24) & 0xF, t2 = rec >> 28;
        const float margin = m->devpair_bound[2 * k], rb = m->devpair_bound[2 * k + 1];
        int pass = 1;
        if (!surv) {
          float dx[3] = {c->gxpos[3 * g2] - c->gxpos[3 * g1], c->gxpos[3 * g2 + 1] - c->gxpos[3 * g1 + 1], c->gxpos[3 * g2 + 2] - c->gxpos[3 * g1 + 2]};
          if (t1 == 0) {
            float n[3] = {c->gxmat[9 * g1 + 2], c->gxmat[9 * g1 + 5], c->gxmat[9 * g1 + 8]};
            pass = dot3f(dx, n) <= rb + margin;
          } else {
            float r = rb + margin;
            pass = dot3f(dx, dx) <= r * r;
          }
          if (kGate && pass) { const int gi = m->devpair_gate[k]; if (gi >= 0 && GRX_GATE_CLEAR(gi)) pass = 0; }
        }
        pq = m->devpair[k];
        if (pass) {
          const int p = pq;
#ifdef GRX_DBG_NO_OBB
          if (t2 == 7 && t1 != 0) { if (S::kHullFilter) ismesh = 1; }
#else
          if (t2 == 7 && t1 != 0) { if (S::kHullFilter) ismesh = grx_obb_overlap(m, c, g1, g2, margin); }
#endif
          else if (t1 == 2 && t2 == 2) grx_sphere_sphere_raw(c, p, c->gxpos + 3 * g1, m->geom_size[3 * g1], c->gxpos + 3 * g2, m->geom_size[3 * g2], margin);
          else if (t1 == 2 && t2 == 3) grx_sphere_capsule(m, c, p, g1, g2, margin);
          else if (t1 == 0 && t2 == 2) grx_plane_sphere(m, c, p, g1, g2, margin);
          else if (t1 == 0 && t2 == 3) grx_plane_capsule(m, c, p, g1, g2, margin);
          else if (t1 == 3 && t2 == 6) grx_capsule_box(m, c, p, g1, g2, margin);
          else if (t1 == 3 && t2 == 3) grx_capsule_capsule(m, c, p, g1, g2, margin);
          else if (t1 == 2 && t2 == 6) grx_sphere_box(m, c, p, g1, g2, margin);
          else if (t1 == 0 && t2 == 6) grx_plane_box(m, c, p, g1, g2, margin);
          else if (t1 == 6 && t2 == 6) isbox = 1;
          else if (S::kConvex && t1 == 0 && t2 == 4) grx_plane_ellipsoid(m, c, p, g1, g2, margin);
          else if (S::kConvex && t1 == 0 && t2 == 5) grx_plane_cylinder(m, c, p, g1, g2, margin);
          else if (S::kConvex && t1 >= 2 && t2 <= 6 && (t1 == 4 || t1 == 5 || t2 == 4 || t2 == 5)) grx_convex_pair(m, c, p, g1, g2, t1, t2, margin);
          else if (t1 == 0 && t2 == 7) {
            if (m->geom_meshnum[g2] <= 32) grx_plane_mesh_small(m, c, p, g1, g2, margin);
            else { int q = GRX_ATOMIC_ADD(&c->cnt[7], 1); if (q < 32) c->ired[q] = p; }
          }
        }
      }
      LV(boxq) = isbox; LV(meshq) = ismesh; LV(pairq) = pq;
    }
    WAVE_SYNC();
    GRX_SUBTICK(c, 13);
    if (S::kMesh) {   // hull-vs-convex pairs that passed the bounding-box filter: pair order, the whole wave on each
      const unsigned long long mm = GRX_BALLOT(meshq);
      if (__builtin_expect(mm != 0ull, 0)) {   // marked cold: the register allocator then places the spill code this region needs around IT instead of inside the hot stages
        if (c->handoff != nullptr && c->soft_maxefc > 0) { LANE0 { c->cnt[2] |= GRX_ST_SOFT; } }   // launches that serve as the standing lane of a hull-less fast kernel: hull activity keeps the world in the lane (grx_lane_ticket)
        int* queue = (int*)(c->Jp + 128);   // the Jacobian pool is free until the constraint stage; [0, 128) is c->red
        FOR_LANES { if (LV(meshq)) queue[__builtin_popcountll(mm & ((1ull << lane) - 1ull))] = LV(pairq); }
        WAVE_SYNC();
#ifndef GRX_DBG_NO_MESHPAIRS
        grx_mesh_pairs(m, c, queue, __builtin_popcountll(mm), lane_);
#endif
      }
    }
    if (S::kHandoff) {   // no hull routine in this kernel: a pair that passed the filters ends the substep here, the world is handed off (grx_forward_euler returns, grx_lane_handoff)
      if (GRX_BALLOT(meshq) != 0ull) {
        if (c->bail == 1 && c->handoff != nullptr) { LANE0 { c->cnt[2] |= GRX_ST_HULL; } WAVE_SYNC(); return; }
        LANE0 { c->cnt[2] |= GRX_ST_CON_OVERFLOW; }   // no lane to hand it to (entry list full, or a launch without one): the pair is NOT collided, and the sticky flag says so
      }
    }
    GRX_SUBTICK(c, 16);
    // box-box pairs that passed the broad phase: queue them (pair order) and let eight lanes work on each
    {
      const unsigned long long bm = GRX_BALLOT(boxq);
      if (bm) {
        int* queue = (int*)c->red;
        FOR_LANES { if (LV(boxq)) queue[__builtin_popcountll(bm & ((1ull << lane) - 1ull))] = LV(pairq); }
        WAVE_SYNC();
        grx_box_box_queue(m, c, queue, __builtin_popcountll(bm), lane_);
      }
    }
    GRX_SUBTICK(c, 11);
    // large hulls (a moving link near the plane): all lanes scan the vertices of one pair at a time
    int nbig = c->cnt[7] < 32 ? c->cnt[7] : 32;
    for (int l = 0; l < nbig; l++) {
      int p = c->ired[l];
      int g1 = m->pair_geom1[p], g2 = m->pair_geom2[p];
      int adr = m->geom_meshadr[g2], num = m->geom_meshnum[g2];
      float margin = m->pair_margin[p];
      const float* gm = c->gxmat + 9 * g2;
      float n[3] = {c->gxmat[9 * g1 + 2], c->gxmat[9 * g1 + 5], c->gxmat[9 * g1 + 8]}, nl[3];
      mulMatTVec3f(nl, gm, n);
      float off = dot3f(c->gxpos + 3 * g2, n) - dot3f(c->gxpos + 3 * g1, n);
      FOR_LANES {
        float bd = 1e30f; int bi = -1;
        for (int v = lane; v < num; v += 64) {
          float dd = m->mesh_vert[3 * (adr + v)] * nl[0] + m->mesh_vert[3 * (adr + v) + 1] * nl[1] + m->mesh_vert[3 * (adr + v) + 2] * nl[2] + off;
          if (dd < bd) { bd = dd; bi = v; }
        }
        c->red[lane] = bd; c->red[64 + lane] = (float)bi;
      }
      WAVE_SYNC();
      LANE0 {
        float bd = 1e30f; int best = -1;
        for (int e = 0; e < 64; e++) {
          float dd = c->red[e]; int vi = (int)c->red[64 + e];
          if (vi >= 0 && (dd < bd || (dd == bd && vi < best))) { bd = dd; best = vi; }
        }
        if (best >= 0 && bd <= margin) {
          int aa = m->mesh_adjadr[adr + best], an = m->mesh_adjnum[adr + best], cn = 0;
          for (int e = -1; e < an && cn < 4; e++) {
            int v = (e < 0) ? best : m->mesh_adj[aa + e];
            float lv[3] = {m->mesh_vert[3 * (adr + v)], m->mesh_vert[3 * (adr + v) + 1], m->mesh_vert[3 * (adr + v) + 2]}, w[3], pos[3];
            float dd = lv[0] * nl[0] + lv[1] * nl[1] + lv[2] * nl[2] + off;
            if (e >= 0 && dd > margin) continue;
            mulMatVec3f(w, gm, lv);
            for (int t = 0; t < 3; t++) pos[t] = w[t] + c->gxpos[3 * g2 + t] - 0.5f * dd * n[t];
            grx_add_contact(c, p, pos, n, dd); cn++;
          }
        }
        c->cnt[7] = 0;
      }
      WAVE_SYNC();
    }
  }
  c0 += chunk;
  } while (kChunked && c0 < ncand);
  LANE0 { if (c->cnt[0] > c->maxcon) c->cnt[0] = c->maxcon; }
  WAVE_SYNC();
  // The noslip sweeps are Gauss-Seidel over the contact list: while they have not converged their iterates depend on the ORDER of the list.
  // The late queues above (box-box, hull pairs, large plane-mesh pairs) append their contacts after everything else; put the list back into
  // pair order (stable: a pair's contacts keep their order), the order of the reference's list.  One lane per contact, rank by counting.
  if ((S::kNoslip && m->noslip_iterations > 0) || classed) {
    const int nc = c->cnt[0];
    GRX_LANEVAR_I(rk); GRX_LANEVAR_I(pk); GRX_LANEVAR(dk); GRX_LANEVAR(x0); GRX_LANEVAR(x1); GRX_LANEVAR(x2); GRX_LANEVAR(f0); GRX_LANEVAR(f1); GRX_LANEVAR(f2);
    FOR_LANES {
      int r = 0, key = 0;
      if (lane < nc) {
        key = c->con_pair[lane];
        for (int j = 0; j < nc; j++) { const int kj = c->con_pair[j]; r += (kj < key) || (kj == key && j < lane); }
        LV(dk) = c->con_dist[lane];
        LV(x0) = c->con_pos[3 * lane]; LV(x1) = c->con_pos[3 * lane + 1]; LV(x2) = c->con_pos[3 * lane + 2];
        LV(f0) = c->con_frame[3 * lane]; LV(f1) = c->con_frame[3 * lane + 1]; LV(f2) = c->con_frame[3 * lane + 2];
      }
      LV(rk) = r; LV(pk) = key;
    }
    WAVE_SYNC();
    FOR_LANES {
      if (lane < nc) {
        const int r = LV(rk);
        c->con_pair[r] = LV(pk); c->con_dist[r] = LV(dk);
        c->con_pos[3 * r] = LV(x0); c->con_pos[3 * r + 1] = LV(x1); c->con_pos[3 * r + 2] = LV(x2);
        c->con_frame[3 * r] = LV(f0); c->con_frame[3 * r + 1] = LV(f1); c->con_frame[3 * r + 2] = LV(f2);
      }
    }
    WAVE_SYNC();
  }
}

