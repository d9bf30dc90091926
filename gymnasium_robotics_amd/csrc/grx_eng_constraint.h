// grx_eng_constraint.h -- K9: constraint rows (welds, joint equalities, friction loss, limits, pyramidal contacts), impedance / aref / R, packed Jacobian rows.
// A FRAGMENT of csrc/grx_engine.h: textually included INSIDE `template <class S> struct GrxEngine { ... }` (every function here is a static member), in the order the engine
// header lists; not a standalone header.  The split is purely textual (round 5): the token stream of the translation units is unchanged.
// ------------------------------------------------------------------------------------------
// K9 constraint rows (equality weld, dof frictionloss, joint limits, pyramidal contacts)
// ------------------------------------------------------------------------------------------
GRX_MEM float grx_impedance(const float* solimp, float pos) {
  float dmin = fminf(GRX_MAXIMP, fmaxf(GRX_MINIMP, solimp[0])), dmax = fminf(GRX_MAXIMP, fmaxf(GRX_MINIMP, solimp[1]));
  float width = fmaxf(0.0f, solimp[2]), mid = fminf(GRX_MAXIMP, fmaxf(GRX_MINIMP, solimp[3])), power = fmaxf(1.0f, solimp[4]);
  if (dmin == dmax || width <= GRX_MINVAL) return 0.5f * (dmin + dmax);
  float x = fabsf(pos) / width;
  if (x >= 1) return dmax;
  if (x <= 0) return dmin;
  float y;
  if (power == 1.0f) y = x;
  else if (power == 2.0f) y = (x <= mid) ? x * x / mid : 1.0f - (1.0f - x) * (1.0f - x) / (1.0f - mid);
  else if (x <= mid) y = powf(x, power) / powf(mid, power - 1.0f);
  else y = 1.0f - powf(1.0f - x, power) / powf(1.0f - mid, power - 1.0f);
  return dmin + y * (dmax - dmin);
}

// column d of the translational / rotational Jacobian of a world point on body b (zero if d not in chain)
// (the body's dof-chain mask and tree root come from its record -- GrxModel::reci_chain, or the weld's own record -- loaded by the caller in its one volley of table reads)
GRX_MEM void grx_jac_col(const GrxCtx* c, unsigned lo, unsigned hi, int rootid, const float* point, int d, float* jp, float* jr) {
  int in = d < 32 ? (lo >> d) & 1u : (hi >> (d - 32)) & 1u;
  if (!in) { jp[0] = jp[1] = jp[2] = 0; jr[0] = jr[1] = jr[2] = 0; return; }
  const float* cref = c->xpos + 3 * rootid;
  float off[3] = {point[0] - cref[0], point[1] - cref[1], point[2] - cref[2]}, t[3];
  const float* cd = c->cdof + 6 * d;
  float w[3] = {cd[0], cd[1], cd[2]};
  cross3f(t, w, off);
  jr[0] = w[0]; jr[1] = w[1]; jr[2] = w[2]; jp[0] = cd[3] + t[0]; jp[1] = cd[4] + t[1]; jp[2] = cd[5] + t[2];
}

// dof span [lo, lo+len) of a 64-bit dof mask
GRX_MEM void grx_mask_span(unsigned long long msk, int* lo, int* len) {
  if (!msk) { *lo = 0; *len = 0; return; }
  int l = __builtin_ctzll(msk), h = 63 - __builtin_clzll(msk);
  *lo = l; *len = h - l + 1;
}
GRX_MEM unsigned long long grx_chainmask(const GrxModel* m, int b) {
  return ((unsigned long long)(unsigned)m->dof_chainmask[2 * b + 1] << 32) | (unsigned)m->dof_chainmask[2 * b];
}
// efc_row[r] = off | lo << 14 | len << 21: 14-bit pool offsets (the large tables of the overflow lane hold up to 16 368 words), dof spans below 128
#define GRX_ROW_OFF(info) ((info) & 0x3FFF)
#define GRX_ROW_LO(info) (((info) >> 14) & 0x7F)
#define GRX_ROW_LEN(info) (((info) >> 21) & 0x7F)
#define GRX_ROW_PACK(off, lo, len) ((off) | ((lo) << 14) | ((len) << 21))
#define GRX_ROW_FROM_STATIC(x) GRX_ROW_PACK((x) & 0xFFF, ((x) >> 12) & 0xFF, ((x) >> 20) & 0xFF)   // the compiler's static rows (weld_row, jeq_row): off | lo << 12 | len << 20
// Second dof span of a row (contacts whose two body chains leave a gap of unused dofs between them): it rides in the upper bits of
// efc_id = sub | id << 4 | loB << 12 | lenB << 20, and its entries follow the first span's entries in the pool.
#define GRX_ROW_IDOF(id) (((id) >> 4) & 0xFF)
#define GRX_ROWB_LO(id) (((id) >> 12) & 0xFF)
#define GRX_ROWB_LEN(id) (((id) >> 20) & 0xFF)
// index of dof d inside the row's storage, or -1 when the row has no entry for it
GRX_MEM int grx_row_pos(int info, int id, int d) {
  const int ja = d - GRX_ROW_LO(info);
  if ((unsigned)ja < (unsigned)GRX_ROW_LEN(info)) return ja;
  if (!S::kTwoSpan) return -1;
  const int jb = d - GRX_ROWB_LO(id);
  return ((unsigned)jb < (unsigned)GRX_ROWB_LEN(id)) ? GRX_ROW_LEN(info) + jb : -1;
}
// row r of J times a dof vector
GRX_MEM float grx_row_dot(const GrxCtx* c, int r, const float* v) {
  const int info = c->efc_row[r], off = GRX_ROW_OFF(info), lo = GRX_ROW_LO(info), len = GRX_ROW_LEN(info);
  float s = 0;
#pragma unroll 8
  for (int j = 0; j < len; j++) s += c->Jp[off + j] * v[lo + j];
  if (S::kTwoSpan) {
    const int id = c->efc_id[r], lob = GRX_ROWB_LO(id), lenb = GRX_ROWB_LEN(id);
#pragma unroll 2
    for (int j = 0; j < lenb; j++) s += c->Jp[off + len + j] * v[lob + j];
  }
  return s;
}

GRX_MEM void grx_make_constraint(const GrxModel* m, GrxCtx* c, int lane_) {
  // The lane index is made opaque for this stage (GRX_OPAQUE_LANE, grx_engine.h): its cheap lane-derived values (packed row descriptors) are recomputed here.
  // Doing this for the whole pass costs more recomputation than it saves (measured -3 % on the hand models).
  GRX_OPAQUE_LANE(lane_);
  GRX_FRESH_MODEL(m, c);
  const int nv = GRX_NVC;
  const int ncon = c->cnt[0];
  // ---- row bookkeeping: one lane per joint (limit flags) and one lane per contact (row count, dof span), then
  // exclusive prefix sums across the wave give every limit / contact its first row and its Jacobian-pool offset.
  const int nwr = 6 * m->nweld, ne = nwr + m->njeq, nf = m->nfric, wpool = m->wpool;   // equality rows: the welds' six each, then one per joint equality
  GRX_LANEVAR_I(limc); GRX_LANEVAR_I(conr); GRX_LANEVAR_I(conw); GRX_LANEVAR_I(coni);
  GRX_LANEVAR_I(tenf); GRX_LANEVAR_I(tenc); GRX_LANEVAR_I(tenw); GRX_LANEVAR(tenl);
  GRX_LANEVAR_I(jdd); GRX_LANEVAR(jq); GRX_LANEVAR(jr0); GRX_LANEVAR(jr1);   // the joint's lane keeps what its limit rows need: dof, joint value, range
  FOR_LANES {
    int f = 0;
    LV(jdd) = 0; LV(jq) = 0.0f; LV(jr0) = 0.0f; LV(jr1) = 0.0f;
    if (lane < GRX_NJC) {
      const int j = lane;
      const int* JI = m->reci_jnt + GRX_RJI * j; const float* JF = m->recf_jnt + GRX_RJF * j;   // one record per joint (GrxModel::reci_jnt): no table walk
      const int lim = JI[6], qa = JI[0], dd = JI[3]; const float mg = JF[12 + GRX_PRM_MARGIN], r0 = JF[8], r1 = JF[9];
      if (lim) {
        float q = c->qpos[qa];
        if (q - r0 < mg) f |= 1;
        if (r1 - q < mg) f |= 2;
        LV(jq) = q;
      }
      LV(jdd) = dd; LV(jr0) = r0; LV(jr1) = r1;
      c->ired[j] = f;
    }
    LV(limc) = (f & 1) + ((f >> 1) & 1);
    int nr = 0, slen = 0;
    if (lane < ncon) {
      const int k = lane;
      const int p = c->con_pair[k];
      const int* PI = m->reci_pair + GRX_RPI * p; const float* PF = m->recf_pair + GRX_RPF * p;   // one record per candidate pair (GrxModel::reci_pair)
      const int dim = PI[0], cb1 = PI[3], cb2 = PI[4], sp = PI[5];   // sp, static: the two dof spans of the pair's body chains
      int active = c->con_dist[k] < PF[2];   // margin - gap
      nr = active ? ((dim == 1) ? 1 : 2 * (dim - 1)) : 0;
      c->con_b1[k] = cb1; c->con_b2[k] = cb2;
      slen = ((sp >> 8) & 0xFF) + ((sp >> 24) & 0xFF);
      c->con_span[k] = sp;
    }
    LV(conr) = nr; LV(conw) = nr * slen; LV(coni) = nr ? slen : 0;
    // fixed-tendon limits: one lane per tendon (length = sum coef * qpos)
    int tf = 0; float tl = 0.0f;
    if (lane < m->ntendon && m->tendon_limited[lane]) {
      const int t = lane;
      for (int w = m->tendon_adr[t]; w < m->tendon_adr[t] + m->tendon_num[t]; w++) tl += m->wrap_coef[w] * c->qpos[m->wrap_qadr[w]];
      const float mg = m->tendon_margin[t];
      if (tl - m->tendon_range[2 * t] < mg) tf |= 1;
      if (m->tendon_range[2 * t + 1] - tl < mg) tf |= 2;
    }
    LV(tenf) = tf; LV(tenl) = tl;
    LV(tenc) = (tf & 1) + ((tf >> 1) & 1);
    LV(tenw) = LV(tenc) * (lane < m->ntendon ? (m->tendon_span[lane] >> 8) : 0);
  }
  WAVE_SYNC();
  GRX_SUBTICK(c, 0);
  GRX_LANEVAR_I(limx); GRX_LANEVAR_I(conrx); GRX_LANEVAR_I(conwx); GRX_LANEVAR_I(tenx); GRX_LANEVAR_I(tenwx);
  int nl, nc_all, pool_all, nlt = 0, tpool = 0;
  GRX_SCAN_EXCL(limc, limx, nl);
  if (m->ntendon) { GRX_SCAN_EXCL(tenc, tenx, nlt); GRX_SCAN_EXCL(tenw, tenwx, tpool); }
  const int nlj = nl;   // joint-limit rows; tendon-limit rows follow them (MuJoCo's row order)
  nl += nlt;
  GRX_SCAN_EXCL(conr, conrx, nc_all);
  GRX_SCAN_EXCL(conw, conwx, pool_all);
  // contacts come last: keep as many whole contacts as fit into the row table and the Jacobian pool
  const int rows0 = ne + nf + nl, pool0 = wpool + nf + nlj + tpool;
  const int maxefc = c->maxefc, jpool = c->jpool;
  int overflow = (rows0 > maxefc) || (pool0 > jpool), ncon_fit = ncon, nc = nc_all;
  GRX_PMAX(c, 32, rows0 + nc_all); GRX_PMAX(c, 33, pool0 + pool_all); GRX_PMAX(c, 34, c->cnt[0]);
  if (c->soft_maxefc > 0 && (rows0 + nc_all > c->soft_maxefc || pool0 + pool_all > c->soft_jpool || c->cnt[0] > c->soft_maxcon)) { LANE0 { c->cnt[2] |= GRX_ST_SOFT; } }
  if (rows0 + nc_all > maxefc || pool0 + pool_all > jpool) {  // rare: find the first contact that does not fit
    GRX_LANEVAR(failp);
    FOR_LANES {
      int fits = (lane >= ncon) || (rows0 + LV(conrx) + LV(conr) <= maxefc && pool0 + LV(conwx) + LV(conw) <= jpool);
      LV(failp) = fits ? -1000.0f : -(float)lane;
    }
    const float mx = grx_reduce_max(failp);
    if (mx > -999.0f) { ncon_fit = (int)(-mx); overflow = 1; nc = GRX_LANE_READ_I(conrx, ncon_fit); }
  }
  int nefc = rows0 + nc;
  if (nefc > maxefc) nefc = maxefc;
  // items of the contact-Jacobian pass: one per (kept contact, dof of its spans); the running item offset (con_ioff) lets an
  // item find its contact with a binary search
  GRX_LANEVAR_I(conix); int nitem;
  FOR_LANES { if (lane >= ncon_fit) LV(coni) = 0; }
  GRX_SCAN_EXCL(coni, conix, nitem);
  GRX_SUBTICK(c, 1);
  // ---- descriptors
  FOR_LANES {
    if (lane < nwr) {  // welds: spans and pool offsets are static (weld_row)
      const int r = lane, w = r / 6, sub = r - 6 * w, info0 = GRX_ROW_FROM_STATIC(m->weld_row[w]);
      c->efc_kind[r] = GRX_ROW_EQ; c->efc_id[r] = (m->weld_eq[w] << 4) | sub;
      c->efc_row[r] = info0 + sub * GRX_ROW_LEN(info0);  // the offset field is the low one: adding sub*len moves to row sub
    } else if (lane < ne) {  // joint equalities (sub 8: their invweight sits in the second eq_invweight slot too)
      const int r = lane, j = r - nwr;
      c->efc_kind[r] = GRX_ROW_EQ; c->efc_id[r] = (m->jeq_eq[j] << 4) | 8; c->efc_row[r] = GRX_ROW_FROM_STATIC(m->jeq_row[j]);
    }
    if ((S::kFixed ? S::NF > 0 : true) && nf > 0)   // compile-time dead for the shapes without friction-loss dofs
      for (int d = lane; d < nv; d += 64) {
        if (m->dof_frictionloss[d] > 0) {
          int r = ne; for (int q = 0; q < d; q++) if (m->dof_frictionloss[q] > 0) r++;
          c->efc_kind[r] = GRX_ROW_FRICTION; c->efc_id[r] = d << 4; c->efc_row[r] = GRX_ROW_PACK(wpool + (r - ne), d, 1);
        }
      }
    if (lane < GRX_NJC) {   // joint limits: the joint's lane writes the whole row -- descriptor, the single Jacobian entry and the residual (it holds the joint value and the range)
      const int j = lane, f = c->ired[j];
      if (f) {
        int r = ne + nf + LV(limx);
        const int dd = LV(jdd); const float q = LV(jq);
        if (f & 1) { if (r < nefc) { c->efc_kind[r] = GRX_ROW_LIMIT; c->efc_id[r] = j << 4; c->efc_row[r] = GRX_ROW_PACK(wpool + (r - ne), dd, 1); c->Jp[wpool + (r - ne)] = 1.0f; c->efc_pos[r] = q - LV(jr0); } r++; }
        if (f & 2) { if (r < nefc) { c->efc_kind[r] = GRX_ROW_LIMIT; c->efc_id[r] = (j << 4) | 1; c->efc_row[r] = GRX_ROW_PACK(wpool + (r - ne), dd, 1); c->Jp[wpool + (r - ne)] = -1.0f; c->efc_pos[r] = LV(jr1) - q; } }
      }
    }
    if (LV(tenf)) {
      const int t = lane, f = LV(tenf), sp = m->tendon_span[t], slo = sp & 0xFF, slen = sp >> 8;
      int r = ne + nf + nlj + LV(tenx), off = wpool + nf + nlj + LV(tenwx);
      if (f & 1) { if (r < nefc) { c->efc_kind[r] = GRX_ROW_TENDON; c->efc_id[r] = t << 4; c->efc_row[r] = GRX_ROW_PACK(off, slo, slen); } r++; off += slen; }
      if (f & 2) { if (r < nefc) { c->efc_kind[r] = GRX_ROW_TENDON; c->efc_id[r] = (t << 4) | 1; c->efc_row[r] = GRX_ROW_PACK(off, slo, slen); } }
    }
    if (lane < ncon) {
      const int k = lane;
      int nr = (k < ncon_fit) ? LV(conr) : 0;
      int r = rows0 + LV(conrx), off = pool0 + LV(conwx);
      c->con_efc[k] = nr ? r : -1;
      c->con_nr[k] = nr;
      const int sp = c->con_span[k], slo = sp & 0xFF, slena = (sp >> 8) & 0xFF, slob = (sp >> 16) & 0xFF, slenb = (sp >> 24) & 0xFF, slen = slena + slenb;
      c->con_ioff[k] = LV(conix);
      for (int q = 0; q < nr; q++) {
        c->efc_kind[r + q] = GRX_ROW_CONTACT; c->efc_id[r + q] = (k << 4) | q | (slob << 12) | (slenb << 20);
        c->efc_row[r + q] = GRX_ROW_PACK(off + q * slen, slo, slena);
      }
    }
  }
  LANE0 { c->cnt[1] = nefc; c->cnt[3] = ne; c->cnt[4] = nf; c->cnt[5] = nl; if (overflow) c->cnt[2] |= GRX_ST_EFC_OVERFLOW; }
  WAVE_SYNC();
  GRX_SUBTICK(c, 2);
  // ---- Jacobian rows.  zero fill, then per (row-group, dof) items
  // (every (row group, dof) item below writes all of its entries, zeros included: no separate clear of J)
  FOR_LANES {
    // welds: one lane per (weld, dof)
    for (int it = lane; it < (nwr / 6) * nv; it += 64) {
      int w = it / nv, d = it - w * nv;
      const int* WI = m->reci_weld + GRX_RWI * w; const float* data = m->recf_weld + GRX_RWF * w; const float* rel = data + 12;   // one record per weld: bodies, their dof chains, eq_data, eq_relpose
      const int b0 = WI[1], b1 = WI[2];
      const unsigned c0lo = (unsigned)WI[4], c0hi = (unsigned)WI[5], c1lo = (unsigned)WI[7], c1hi = (unsigned)WI[8]; const int root0 = WI[6], root1 = WI[9];
      float bx[2][3], bq[2][4], pos[2][3];
      for (int s = 0; s < 2; s++) {
        int bb = s ? b1 : b0; float v[3], rp[3] = {rel[7 * s], rel[7 * s + 1], rel[7 * s + 2]}, rq[4] = {rel[7 * s + 3], rel[7 * s + 4], rel[7 * s + 5], rel[7 * s + 6]};
        mulMatVec3f(v, c->xmat + 9 * bb, rp);
        for (int k = 0; k < 3; k++) bx[s][k] = c->xpos[3 * bb + k] + v[k];
        mulQuatf(bq[s], c->xquat + 4 * bb, rq); normalize4f(bq[s]);
        float an[3] = {data[3 * (1 - s)], data[3 * (1 - s) + 1], data[3 * (1 - s) + 2]};
        rotVecQuatf(v, an, bq[s]);
        for (int k = 0; k < 3; k++) pos[s][k] = bx[s][k] + v[k];
      }
      float jp0[3], jr0[3], jp1[3], jr1[3];
      grx_jac_col(c, c0lo, c0hi, root0, pos[0], d, jp0, jr0); grx_jac_col(c, c1lo, c1hi, root1, pos[1], d, jp1, jr1);
      float ts = data[10];
      float relq[4] = {data[6], data[7], data[8], data[9]}, quat[4], quat1[4] = {bq[1][0], -bq[1][1], -bq[1][2], -bq[1][3]};
      mulQuatf(quat, bq[0], relq);
      float axis[4] = {0, jr0[0] - jr1[0], jr0[1] - jr1[1], jr0[2] - jr1[2]}, t1[4], t2[4];
      mulQuatf(t1, quat1, axis); mulQuatf(t2, t1, quat);
      { int info = c->efc_row[6 * w], jd = d - GRX_ROW_LO(info), len = GRX_ROW_LEN(info), off = GRX_ROW_OFF(info);
        if ((unsigned)jd < (unsigned)len)
          for (int r = 0; r < 3; r++) { c->Jp[off + r * len + jd] = jp0[r] - jp1[r]; c->Jp[off + (3 + r) * len + jd] = 0.5f * ts * t2[1 + r]; } }
      if (d == 0) {  // residuals (one lane per weld)
        float quat2[4]; mulQuatf(quat2, quat1, quat);
        for (int r = 0; r < 3; r++) { c->efc_pos[6 * w + r] = pos[0][r] - pos[1][r]; c->efc_pos[6 * w + 3 + r] = ts * quat2[1 + r]; }
      }
    }
    // joint equalities: one lane per constraint.  r = (q1 - q1_0) - poly(q2 - q2_0), J = e_dof1 - poly'(q2 - q2_0) e_dof2 (MuJoCo mjEQ_JOINT [3P])
    for (int j = lane; j < m->njeq; j += 64) {
      const int r = nwr + j, e = m->jeq_eq[j], info = c->efc_row[r], off = GRX_ROW_OFF(info), lo = GRX_ROW_LO(info), len = GRX_ROW_LEN(info);
      const float* data = m->eq_data + 11 * e;
      const float x = c->qpos[m->jeq_qadr[2 * j + 1]] - data[6];
      const float poly = data[0] + x * (data[1] + x * (data[2] + x * (data[3] + x * data[4])));
      const float deriv = data[1] + x * (2.0f * data[2] + x * (3.0f * data[3] + x * 4.0f * data[4]));
      for (int k = 0; k < len; k++) c->Jp[off + k] = 0.0f;
      c->Jp[off + m->jeq_dof[2 * j] - lo] = 1.0f;
      c->Jp[off + m->jeq_dof[2 * j + 1] - lo] += -deriv;
      c->efc_pos[r] = (c->qpos[m->jeq_qadr[2 * j]] - data[5]) - poly;
    }
    // frictionloss + limits: one lane per row
    GRX_SUBTICK(c, 3);
    for (int r = ne + lane; r < ne + nf && r < nefc; r += 64) {   // (the limit rows that follow them were written by their joints' lanes above)
      c->Jp[GRX_ROW_OFF(c->efc_row[r])] = 1.0f;
      c->efc_pos[r] = 0;
    }
    // tendon limits: one lane per tendon writes its (up to two) rows over the tendon's dof span
    if (LV(tenf)) {
      const int t = lane, f = LV(tenf);
      int r = ne + nf + nlj + LV(tenx);
      for (int side = 0; side < 2; side++) {
        if (!((f >> side) & 1)) continue;
        if (r < nefc) {
          const int info = c->efc_row[r], off = GRX_ROW_OFF(info), lo = GRX_ROW_LO(info), len = GRX_ROW_LEN(info);
          for (int j = 0; j < len; j++) c->Jp[off + j] = 0.0f;
          for (int w = m->tendon_adr[t]; w < m->tendon_adr[t] + m->tendon_num[t]; w++) c->Jp[off + m->wrap_dof[w] - lo] += side ? -m->wrap_coef[w] : m->wrap_coef[w];
          c->efc_pos[r] = side ? m->tendon_range[2 * t + 1] - LV(tenl) : LV(tenl) - m->tendon_range[2 * t];
        }
        r++;
      }
    }
    // contacts: one lane per (contact, dof of its span)
    GRX_SUBTICK(c, 4);
    for (int it = lane; it < nitem; it += 64) {
      int k = 0;  // largest k with item offset <= it (contacts without items share the offset of the next one)
      for (int step = GRX_MAXCON / 2; step > 0; step >>= 1) { int kk = k + step; if (kk < ncon_fit && c->con_ioff[kk] <= it) k = kk; }
      const int jd = it - c->con_ioff[k];
      int r0 = c->con_efc[k];
      const int sp = c->con_span[k], slena = (sp >> 8) & 0xFF, slen = slena + ((sp >> 24) & 0xFF);
      int d = jd < slena ? (sp & 0xFF) + jd : ((sp >> 16) & 0xFF) + jd - slena;
      int p = c->con_pair[k], nrk = c->con_nr[k], dim = (nrk == 1) ? 1 : nrk / 2 + 1;
      int b1 = c->con_b1[k], b2 = c->con_b2[k];
      // one volley of table reads per item: the two bodies' chain records and the pair's friction coefficients
      const int* C1 = m->reci_chain + GRX_RCI * b1; const int* C2 = m->reci_chain + GRX_RCI * b2; const float* PFr = m->recf_pair + GRX_RPF * p + 3;
      const unsigned c1lo = (unsigned)C1[0], c1hi = (unsigned)C1[1], c2lo = (unsigned)C2[0], c2hi = (unsigned)C2[1]; const int root1 = C1[2], root2 = C2[2];
      const float mu0 = PFr[0], mu1 = PFr[1], mu2 = PFr[2], mu3 = PFr[3], mu4 = PFr[4];
      float pos[3] = {c->con_pos[3 * k], c->con_pos[3 * k + 1], c->con_pos[3 * k + 2]};
      float jp1[3], jr1[3], jp2[3], jr2[3];
      grx_jac_col(c, c1lo, c1hi, root1, pos, d, jp1, jr1); grx_jac_col(c, c2lo, c2hi, root2, pos, d, jp2, jr2);
      float dp[3] = {jp2[0] - jp1[0], jp2[1] - jp1[1], jp2[2] - jp1[2]}, dr[3] = {jr2[0] - jr1[0], jr2[1] - jr1[1], jr2[2] - jr1[2]};
      float fr[9] = {c->con_frame[3 * k], c->con_frame[3 * k + 1], c->con_frame[3 * k + 2], 0, 0, 0, 0, 0, 0};
      grx_make_frame(fr);
      float jc[6];
      for (int r = 0; r < 3; r++) { jc[r] = fr[3 * r] * dp[0] + fr[3 * r + 1] * dp[1] + fr[3 * r + 2] * dp[2]; jc[3 + r] = fr[3 * r] * dr[0] + fr[3 * r + 1] * dr[1] + fr[3 * r + 2] * dr[2]; }
      const int off0 = GRX_ROW_OFF(c->efc_row[r0]);
      if (dim == 1) c->Jp[off0 + jd] = jc[0];
      else {
#pragma unroll
        for (int q = 1; q < 6; q++) {   // (condim <= 6; unrolled: the coefficient and jc[q] are registers, not a dynamically indexed array)
          if (q < dim) {
            const float mu = q == 1 ? mu0 : (q == 2 ? mu1 : (q == 3 ? mu2 : (q == 4 ? mu3 : mu4)));
            int ro = off0 + 2 * (q - 1) * slen + jd;
            c->Jp[ro] = jc[0] + mu * jc[q];
            c->Jp[ro + slen] = jc[0] - mu * jc[q];
          }
        }
      }
    }
  }
  WAVE_SYNC();
  GRX_SUBTICK(c, 5);
  // ---- per-row impedance, regulariser, reference acceleration
  FOR_LANES {
    for (int r = lane; r < nefc; r += 64) {
      int kind = c->efc_kind[r], id = GRX_ROW_IDOF(c->efc_id[r]), sub = c->efc_id[r] & 15;
      // ONE row-parameter block per row, whatever its kind (GRX_PRM_*: solref, solimp, margin, diagonal approximation, friction / friction loss), at a kind-specific place of the
      // entity's record: one volley of table reads for the whole wave instead of one dependent chain per kind (the kinds' branches ran one after the other)
      const float* P; float pos; int iscon = 0;
      if (kind == GRX_ROW_EQ) { P = m->recf_eq + GRX_REF * id; pos = c->efc_pos[r]; }
      else if (kind == GRX_ROW_FRICTION) { P = m->recf_dof + GRX_RDF * id + 4; pos = 0; }
      else if (kind == GRX_ROW_LIMIT) { P = m->recf_jnt + GRX_RJF * id + 12; pos = c->efc_pos[r]; }
      else if (kind == GRX_ROW_TENDON) { P = m->recf_ten + GRX_RTF * id; pos = c->efc_pos[r]; }
      else { const int p = c->con_pair[id]; P = m->recf_pair + GRX_RPF * p + 8; pos = c->con_dist[id]; iscon = 1; }
      float solref[2] = {P[GRX_PRM_SOLREF], P[GRX_PRM_SOLREF + 1]}, solimp[5] = {P[GRX_PRM_SOLIMP], P[GRX_PRM_SOLIMP + 1], P[GRX_PRM_SOLIMP + 2], P[GRX_PRM_SOLIMP + 3], P[GRX_PRM_SOLIMP + 4]};
      const float margin = P[GRX_PRM_MARGIN], da0 = P[GRX_PRM_DA], aux = P[GRX_PRM_AUX], da2 = P[GRX_PRM_DA2];
      float dA = da0, floss = 0, rscale = 1.0f;
      if (kind == GRX_ROW_EQ) dA = (sub >= 3) ? da2 : da0;
      else if (kind == GRX_ROW_FRICTION) floss = aux;
      else if (iscon) {
        const float tran = da0;
        if (da2 == 1.0f) dA = tran;   // condim 1 (the pair record keeps condim in the block's spare slot)
        else {  // every pyramid row of a contact shares R = 2 mu^2 R(first row)
          float f0 = aux;
          dA = tran + f0 * f0 * tran;
          float mu = f0 / sqrtf(m->impratio);
          rscale = 2.0f * mu * mu;
        }
        c->efc_pos[r] = pos;
      }
      float imp = grx_impedance(solimp, pos - margin);
      float dmax = fminf(GRX_MAXIMP, fmaxf(GRX_MINIMP, solimp[1]));
      float kk, bb;
      if (solref[0] > 0) { float tc = fmaxf(solref[0], 2.0f * m->timestep), dr = solref[1]; kk = 1.0f / (dmax * dmax * tc * tc * dr * dr); bb = 2.0f / (dmax * tc); }
      else { kk = -solref[0] / (dmax * dmax); bb = -solref[1] / dmax; }
      if (kind == GRX_ROW_FRICTION) kk = 0;
      float R = fmaxf(GRX_MINVAL, (1.0f - imp) * dA / imp) * rscale;
      const float vel = grx_row_dot(c, r, c->qvel);
      c->efc_D[r] = 1.0f / R;
      c->efc_aref[r] = -bb * vel - kk * imp * (pos - margin);
      if (m->nfric) c->efc_floss[r] = floss;
    }
  }
  WAVE_SYNC();
}

