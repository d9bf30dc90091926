// grx_host_model.h -- host-side packing of the compiled-model blob (include/grx_model_fields.def)
// into one fp32 and one int32 array, and a GrxModel whose table pointers address those arrays
// relative to arbitrary base pointers (host memory for the lane emulator, HBM for the GPU).
#pragma once
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/grx_model.h"
#include "grx_engine.h"

struct GrxPackedModel {
  std::vector<float> f;       // all GRX_FF tables, converted to fp32
  std::vector<int32_t> i;     // all GRX_FI tables
  std::vector<int> off, cnt;  // per table (in .def order): offset inside f or i, element count
  std::vector<char> kind;     // 'f' or 'i'
  std::vector<std::string> name;
  int off_mesh_nbr = -1;      // offset of the derived neighbour-record table inside f (-1: the model has no hulls)
  int off_cellhdr = -1, off_cellbase = -1, off_cellrec = -1;   // derived support-candidate lists (GrxModel::mesh_cellhdr / geom_cellbase inside i, mesh_cellrec inside f; -1: none)
  // derived per-stage record tables (GrxModel::reci_* inside i, recf_* inside f; grx_build_records): offsets, and the [begin, end) ranges they occupy (re-uploaded after grx_model_set_table)
  struct { int i_body, i_jnt, i_pair, i_chain, i_weld, i_act, i_dof, i_mpair, f_body, f_jnt, f_pair, f_weld, f_eq, f_act, f_dof, f_ten, f_mpair; } rec;
  int rec_i0 = 0, rec_i1 = 0, rec_f0 = 0, rec_f1 = 0;
  GrxModel proto;             // scalar members filled; pointers unset
};
inline void grx_build_records(GrxPackedModel* out, bool allocate);

// Support-candidate lists of one hull (GrxModel::mesh_cellhdr): for every cube-map cell of directions the vertices that can be the support vertex, or tie with it inside the
// device scan's band, for SOME direction of the cell.  With w the support vertex of the cell's centre direction dc and rho >= |d - dc| for every unit d of the (dilated) cell:
// a vertex v with v . d >= max_u u . d - band has (w - v) . d <= band, hence (w - v) . dc <= band + rho |w - v|; everything else is left out.  band = 5e-6 m covers the scan's
// tie band (1e-6 max(1, |t|)), the rounding of its fp32 projections and a direction that is unit only to fp32; the cells are dilated by 1e-3 in cube-map coordinates, a thousand
// times the rounding of grx_hull_cell's index arithmetic (and it makes the lists of neighbouring faces overlap where the major axis is a tie).
inline void grx_build_hull_cells(const double* vert, int num, std::vector<int32_t>* hdr, std::vector<float>* rec) {
  const int G = GRX_CELL_G;
  const double band = 5.0e-6, dil = 1.0e-3;
  std::vector<int> list;
  for (int face = 0; face < 6; face++)
    for (int iu = 0; iu < G; iu++)
      for (int iv = 0; iv < G; iv++) {
        auto dir = [&](double u, double v, double* d) {
          double x[3];
          const double sgn = (face & 1) ? -1.0 : 1.0;
          if (face < 2) { x[0] = sgn; x[1] = u; x[2] = v; } else if (face < 4) { x[0] = u; x[1] = sgn; x[2] = v; } else { x[0] = u; x[1] = v; x[2] = sgn; }
          const double n = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
          d[0] = x[0] / n; d[1] = x[1] / n; d[2] = x[2] / n;
        };
        // grx_hull_cell: u / |major| in [-1, 1] is cut into G equal steps
        const double u0 = -1.0 + 2.0 * iu / G - dil, u1 = -1.0 + 2.0 * (iu + 1) / G + dil, v0 = -1.0 + 2.0 * iv / G - dil, v1 = -1.0 + 2.0 * (iv + 1) / G + dil;
        double dc[3], dk[3], rho = 0.0;
        dir(0.5 * (u0 + u1), 0.5 * (v0 + v1), dc);
        const double cu[4] = {u0, u0, u1, u1}, cv[4] = {v0, v1, v0, v1};
        for (int k = 0; k < 4; k++) {   // the patch is the central projection of a square: its farthest point from the centre direction is a corner
          dir(cu[k], cv[k], dk);
          const double e = std::sqrt((dk[0] - dc[0]) * (dk[0] - dc[0]) + (dk[1] - dc[1]) * (dk[1] - dc[1]) + (dk[2] - dc[2]) * (dk[2] - dc[2]));
          if (e > rho) rho = e;
        }
        rho *= 1.05;
        int w = 0; double tw = -1e300;
        for (int k = 0; k < num; k++) { const double t = vert[3 * k] * dc[0] + vert[3 * k + 1] * dc[1] + vert[3 * k + 2] * dc[2]; if (t > tw) { tw = t; w = k; } }
        list.clear();
        for (int k = 0; k < num; k++) {
          const double dx = vert[3 * w] - vert[3 * k], dy = vert[3 * w + 1] - vert[3 * k + 1], dz = vert[3 * w + 2] - vert[3 * k + 2];
          if (dx * dc[0] + dy * dc[1] + dz * dc[2] <= band + rho * std::sqrt(dx * dx + dy * dy + dz * dz)) list.push_back(k);
        }
        if ((int)list.size() > GRX_CELL_MAX) { hdr->push_back(0); hdr->push_back(0); continue; }   // (a face seen head-on with more vertices than a wave has lanes: the device scans the hull)
        hdr->push_back((int32_t)(rec->size() / 4)); hdr->push_back((int32_t)list.size());
        for (int k : list) {
          float id; const int32_t ki = k; std::memcpy(&id, &ki, 4);
          rec->push_back((float)vert[3 * k]); rec->push_back((float)vert[3 * k + 1]); rec->push_back((float)vert[3 * k + 2]); rec->push_back(id);
        }
      }
}

inline void grx_pack_model(const int32_t* H, const int32_t* I, const double* F, GrxPackedModel* out) {
  grx_model_view v;
  grx_model_view_init(&v, H, I, F);
  out->f.clear(); out->i.clear(); out->off.clear(); out->cnt.clear(); out->kind.clear(); out->name.clear();
#define GRX_FI(n) out->name.push_back(#n); out->kind.push_back('i'); out->off.push_back((int)out->i.size()); out->cnt.push_back(v.n_##n); \
  out->i.insert(out->i.end(), v.n, v.n + v.n_##n); while (out->i.size() % 4) out->i.push_back(0);
#define GRX_FF(n) out->name.push_back(#n); out->kind.push_back('f'); out->off.push_back((int)out->f.size()); out->cnt.push_back(v.n_##n); \
  for (int k_ = 0; k_ < v.n_##n; k_++) out->f.push_back((float)v.n[k_]); while (out->f.size() % 4) out->f.push_back(0.0f);
#include "../../include/grx_model_fields.def"
#undef GRX_FI
#undef GRX_FF
  // derived table (GrxModel::mesh_nbr): for every hull vertex the vertex and its hull neighbours as (x, y, z, tag) records, 16 per vertex
  // (only for models that HAVE a hull-vs-convex candidate pair: the records are 256 bytes per hull vertex of HBM and upload; a model whose hulls only meet planes never reads them)
  out->off_mesh_nbr = -1;
  bool any_hull_pair = false;
  for (int k = 0; k < v.n_devpair_geoms; k++) { const unsigned rec = (unsigned)v.devpair_geoms[k]; if ((rec >> 28) == 7 && ((rec >> 24) & 0xF) != 0) any_hull_pair = true; }
  if (any_hull_pair && v.n_mesh_vert >= 3 && v.n_mesh_adjadr * 3 == v.n_mesh_vert) {
    const int nvert = v.n_mesh_vert / 3;
    out->off_mesh_nbr = (int)out->f.size();
    out->f.resize(out->f.size() + (size_t)nvert * GRX_NBR_RECS * 4, 0.0f);
    float* T = out->f.data() + out->off_mesh_nbr;
    for (int g = 0; g < nvert; g++) { float* r = T + (size_t)g * GRX_NBR_RECS * 4; for (int k = 0; k < GRX_NBR_RECS; k++) r[4 * k + 3] = -1.0f; }   // unusable until a hull claims the vertex
    for (int gi = 0; gi < v.n_geom_hulladr; gi++) {
      const int adr = v.geom_hulladr[gi], num = v.geom_hullnum[gi];
      if (adr < 0 || num <= 0 || adr + num > nvert) continue;
      for (int lv = 0; lv < num; lv++) {
        const int g = adr + lv, aa = v.mesh_adjadr[g], an = v.mesh_adjnum[g];
        float* r = T + (size_t)g * GRX_NBR_RECS * 4;
        r[0] = (float)v.mesh_vert[3 * g]; r[1] = (float)v.mesh_vert[3 * g + 1]; r[2] = (float)v.mesh_vert[3 * g + 2];
        r[3] = (an >= 1 && an <= GRX_NBR_RECS - 1) ? (float)an : -1.0f;
        for (int k = 0; k < an && k < GRX_NBR_RECS - 1; k++) {
          const int nb = v.mesh_adj[aa + k], gn = adr + nb;
          if (nb < 0 || nb >= num) { r[3] = -1.0f; break; }
          float* q = r + 4 * (k + 1);
          q[0] = (float)v.mesh_vert[3 * gn]; q[1] = (float)v.mesh_vert[3 * gn + 1]; q[2] = (float)v.mesh_vert[3 * gn + 2]; q[3] = (float)nb;
        }
      }
    }
  }
  // derived tables (GrxModel::mesh_cellhdr / geom_cellbase / mesh_cellrec): support-candidate lists of the hulls that take part in a hull-vs-convex candidate pair
  out->off_cellhdr = out->off_cellbase = out->off_cellrec = -1;
  if (v.n_mesh_vert >= 3 && v.n_geom_hulladr > 0 && !getenv("GRX_NO_HULLCELLS")) {
    std::vector<char> used(v.n_geom_hulladr, 0);
    for (int k = 0; k < v.n_devpair_geoms; k++) {
      const unsigned rec = (unsigned)v.devpair_geoms[k];
      const int g1 = rec & 0xFFF, g2 = (rec >> 12) & 0xFFF, t1 = (rec >> 24) & 0xF, t2 = rec >> 28;
      if (t2 == 7 && t1 != 0) { if (g2 < v.n_geom_hulladr) used[g2] = 1; if (t1 == 7 && g1 < v.n_geom_hulladr) used[g1] = 1; }
    }
    std::vector<int32_t> hdr, base(v.n_geom_hulladr, -1);
    std::vector<float> rec;
    const int nvert = v.n_mesh_vert / 3;
    for (int gi = 0; gi < v.n_geom_hulladr; gi++) {
      const int adr = v.geom_hulladr[gi], num = v.geom_hullnum[gi];
      if (!used[gi] || adr < 0 || num < GRX_CELL_MAX || adr + num > nvert) continue;   // (a hull of fewer vertices than a wave has lanes is one round of loads anyway)
      for (int gj = 0; gj < gi; gj++) if (base[gj] >= 0 && v.geom_hulladr[gj] == adr && v.geom_hullnum[gj] == num) { base[gi] = base[gj]; break; }   // two geoms of one mesh share the lists
      if (base[gi] >= 0) continue;
      base[gi] = (int32_t)(hdr.size() / 2);
      grx_build_hull_cells(v.mesh_vert + 3 * (size_t)adr, num, &hdr, &rec);
    }
    if (!hdr.empty()) {
      while (out->i.size() % 4) out->i.push_back(0);
      out->off_cellhdr = (int)out->i.size(); out->i.insert(out->i.end(), hdr.begin(), hdr.end());
      while (out->i.size() % 4) out->i.push_back(0);
      out->off_cellbase = (int)out->i.size(); out->i.insert(out->i.end(), base.begin(), base.end());
      while (out->i.size() % 4) out->i.push_back(0);
      while (out->f.size() % 4) out->f.push_back(0.0f);      // 16-byte records
      out->off_cellrec = (int)out->f.size(); out->f.insert(out->f.end(), rec.begin(), rec.end());
    }
  }
  GrxModel& m = out->proto;
  std::memset(&m, 0, sizeof(m));
  const int32_t* d = v.dims;
  m.nq = d[GRX_NQ]; m.nv = d[GRX_NV]; m.nu = d[GRX_NU]; m.nbody = d[GRX_NBODY]; m.njnt = d[GRX_NJNT]; m.ngeom = d[GRX_NGEOM];
  m.nsite = d[GRX_NSITE]; m.nmocap = d[GRX_NMOCAP]; m.neq = d[GRX_NEQ]; m.npair = d[GRX_NPAIR]; m.maxdepth = d[GRX_MAXDEPTH];
  m.eulerdamp = d[GRX_EULERDAMP]; m.ndevpair = v.n_devpair; m.nmpair = v.n_mpair_i;
  m.integrator = d[GRX_INTEGRATOR];
  m.njump = m.nbody > 0 ? v.n_body_jump / m.nbody : 0;
  m.ntendon = v.n_tendon_adr; m.ntouch = v.n_touch_body;
  // trailing free object: the last joint is a free joint of a body hanging off the world (its 6 dofs are the last ones and M has no entries
  // between them and the other dofs)
  m.nfreeobj = 0;
  if (m.njnt > 1 && m.nv > 6 && v.jnt_type[m.njnt - 1] == GRX_JNT_FREE && v.jnt_dofadr[m.njnt - 1] == m.nv - 6 &&
      v.body_parent[v.jnt_bodyid[m.njnt - 1]] == 0)
    m.nfreeobj = 6;
  m.ngridgeom = v.n_grid_geom; m.ngridwall = v.n_grid_wall_geom; m.gridnx = m.gridny = 0; m.gridx0 = m.gridy0 = m.gridinv = 0.0f;
  if (v.n_grid_param >= 5) { m.gridx0 = (float)v.grid_param[0]; m.gridy0 = (float)v.grid_param[1]; m.gridinv = (float)v.grid_param[2]; m.gridnx = (int)v.grid_param[3]; m.gridny = (int)v.grid_param[4]; }
  m.twospan = 0;
  for (int k = 0; k < v.n_pair_span; k++) if (((unsigned)v.pair_span[k] >> 24) != 0) m.twospan = 1;
  // candidate pairs that go to the general convex (MPR) narrow phase: only the generic kernels carry that code
  m.nconvex = 0;
  for (int k = 0; k < v.n_devpair_geoms; k++) {
    const unsigned rec = (unsigned)v.devpair_geoms[k]; const int t1 = (rec >> 24) & 0xF, t2 = rec >> 28;
    if ((t1 == 0 && (t2 == 4 || t2 == 5)) || (t1 >= 2 && t2 <= 6 && (t1 == 4 || t1 == 5 || t2 == 4 || t2 == 5))) m.nconvex++;
  }
  m.nmeshpair = 0;   // hull against a primitive or another hull: the wave-cooperative routine (grx_mesh_pairs) and its 8-word direction cache
  for (int k = 0; k < v.n_devpair_geoms; k++) { const unsigned rec = (unsigned)v.devpair_geoms[k]; if ((rec >> 28) == 7 && ((rec >> 24) & 0xF) != 0) m.nmeshpair++; }
  m.ngate = v.n_gate_qadr / 3 < 64 ? v.n_gate_qadr / 3 : 64;   // one lane per gate (mjcf/pair_gates.py keeps at most 64)
  m.nshift = 0;
  for (int k = 0; k < v.n_body_shift; k++) m.nshift += v.body_shift[k] != 0;
  for (int k = 0; k < v.n_geom_shift; k++) m.nshift += v.geom_shift[k] != 0;
  for (int k = 0; k < v.n_site_shift; k++) m.nshift += v.site_shift[k] != 0;
  m.iterations = d[GRX_ITERATIONS] > 0 ? d[GRX_ITERATIONS] : 100;
  m.noslip_iterations = d[GRX_NOSLIP_ITERATIONS]; m.noslip_tolerance = (float)v.opt[GRX_NOSLIP_TOLERANCE];
  m.nfric = 0; m.nweld = 0; m.wpool = 0;
  for (int k = 0; k < v.n_weld_row; k++) m.wpool += 6 * ((v.weld_row[k] >> 20) & 0xFF);
  m.njeq = v.n_jeq_eq;
  for (int k = 0; k < v.n_jeq_row; k++) m.wpool += (v.jeq_row[k] >> 20) & 0xFF;
  for (int k = 0; k < v.n_dof_frictionloss; k++) if (v.dof_frictionloss[k] > 0) m.nfric++;
  for (int k = 0; k < v.n_eq_type; k++) if (v.eq_active[k] && v.eq_type[k] == GRX_EQ_WELD) m.nweld++;
  // Capacities of the row tables and of the packed-Jacobian pool: the compiler may request more than the defaults for models
  // with wide / tall contact rows (dims slots GRX_MAXEFC_REQ / GRX_JPOOL_REQ, 0 = default); row offsets are 14 bits (GRX_ROW_PACK).
  m.maxefc = d[GRX_MAXEFC_REQ] > 0 ? ((d[GRX_MAXEFC_REQ] + 15) / 16) * 16 : GRX_MAXEFC;
  m.jpool = d[GRX_JPOOL_REQ] > 0 ? ((d[GRX_JPOOL_REQ] + 15) / 16) * 16 : GRX_JPOOL;
  if (m.jpool > 16368) m.jpool = 16368;   // 14-bit row offsets (GRX_ROW_PACK)
  m.maxcon = (d[GRX_MAXCON_REQ] > 0 && d[GRX_MAXCON_REQ] <= GRX_MAXCON) ? d[GRX_MAXCON_REQ] : GRX_MAXCON_DEFAULT;
  {
    static const int hand_parent[24] = GRX_HAND_DOF_PARENTS;
    m.handtree = (v.n_dof_parentid >= 24) ? 1 : 0;
    for (int k = 0; k < 24 && m.handtree; k++) if (v.dof_parentid[k] != hand_parent[k]) m.handtree = 0;
    for (int k = 24; k < v.n_dof_parentid && m.handtree; k++) if (v.dof_parentid[k] >= 0 && v.dof_parentid[k] < 24) m.handtree = 0;   // nothing else hangs off the hand
  }
  m.anydamp = 0;
  for (int k = 0; k < v.n_dof_damping; k++) if (v.dof_damping[k] > 0) m.anydamp = 1;
  m.timestep = (float)v.opt[GRX_TIMESTEP];
  m.gravity[0] = (float)v.opt[GRX_GRAVITY_X]; m.gravity[1] = (float)v.opt[GRX_GRAVITY_Y]; m.gravity[2] = (float)v.opt[GRX_GRAVITY_Z];
  m.meaninertia = (float)v.opt[GRX_MEANINERTIA]; m.impratio = (float)v.opt[GRX_IMPRATIO];
  m.mpr_tolerance = (float)v.opt[GRX_MPR_TOLERANCE]; m.mpr_iterations = (int)v.opt[GRX_MPR_ITERATIONS];
  m.origin[0] = v.opt[GRX_ORIGIN_X]; m.origin[1] = v.opt[GRX_ORIGIN_Y]; m.origin[2] = v.opt[GRX_ORIGIN_Z];
  grx_build_records(out, true);
}

// model view whose tables live at (fbase, ibase)
inline GrxModel grx_bind_model(const GrxPackedModel& p, const float* fbase, const int32_t* ibase) {
  GrxModel m = p.proto;
  int k = 0;
#define GRX_FI(n) m.n = ibase + p.off[k]; ++k;
#define GRX_FF(n) m.n = fbase + p.off[k]; ++k;
#include "../../include/grx_model_fields.def"
#undef GRX_FI
#undef GRX_FF
  m.mesh_nbr = p.off_mesh_nbr >= 0 ? fbase + p.off_mesh_nbr : nullptr;
  m.mesh_cellhdr = p.off_cellhdr >= 0 ? ibase + p.off_cellhdr : nullptr;
  m.geom_cellbase = p.off_cellbase >= 0 ? ibase + p.off_cellbase : nullptr;
  m.mesh_cellrec = p.off_cellrec >= 0 ? fbase + p.off_cellrec : nullptr;
  m.reci_body = ibase + p.rec.i_body; m.reci_jnt = ibase + p.rec.i_jnt; m.reci_pair = ibase + p.rec.i_pair; m.reci_chain = ibase + p.rec.i_chain; m.reci_weld = ibase + p.rec.i_weld;
  m.reci_act = ibase + p.rec.i_act; m.reci_dof = ibase + p.rec.i_dof; m.reci_mpair = ibase + p.rec.i_mpair;
  m.recf_body = fbase + p.rec.f_body; m.recf_jnt = fbase + p.rec.f_jnt; m.recf_pair = fbase + p.rec.f_pair; m.recf_weld = fbase + p.rec.f_weld; m.recf_eq = fbase + p.rec.f_eq;
  m.recf_act = fbase + p.rec.f_act; m.recf_dof = fbase + p.rec.f_dof; m.recf_ten = fbase + p.rec.f_ten; m.recf_mpair = fbase + p.rec.f_mpair;
  return m;
}

inline int grx_find_table(const GrxPackedModel& p, const char* name) {
  for (size_t k = 0; k < p.name.size(); k++) if (p.name[k] == name) return (int)k;
  return -1;
}

// Per-stage records (GrxModel::reci_* / recf_*, layouts: the GRX_R* enums of grx_engine.h).  Built from the PACKED tables (the fp32 values the device reads), so a record field
// is bit for bit what the table walk it replaces would have loaded; the few sums a stage used to form from two table values (margin - gap, the two geoms' invweight0) are formed
// here in the same `float` arithmetic.  allocate: append the (zeroed) regions to the packed arrays first; otherwise refill them in place (after a table edit).
inline void grx_build_records(GrxPackedModel* out, bool allocate) {
  const GrxModel& d = out->proto;
  auto C = [&](const char* n) { const int k = grx_find_table(*out, n); return k >= 0 ? out->cnt[k] : 0; };
  const int nbody = d.nbody, njnt = d.njnt, nv = d.nv, nu = d.nu, npair = C("pair_geom1"), nweld = C("weld_eq"), neq = C("eq_type"), nten = C("tendon_adr"), nmp = C("mpair_i"), njump = d.njump;
  if (allocate) {
    auto AI = [&](int n) { while (out->i.size() % 4) out->i.push_back(0); const int o = (int)out->i.size(); out->i.resize(out->i.size() + (size_t)n, 0); return o; };
    auto AF = [&](int n) { while (out->f.size() % 4) out->f.push_back(0.0f); const int o = (int)out->f.size(); out->f.resize(out->f.size() + (size_t)n, 0.0f); return o; };
    while (out->i.size() % 4) out->i.push_back(0);
    while (out->f.size() % 4) out->f.push_back(0.0f);
    out->rec_i0 = (int)out->i.size(); out->rec_f0 = (int)out->f.size();
    out->rec.i_body = AI(GRX_RBI * nbody); out->rec.i_jnt = AI(GRX_RJI * njnt); out->rec.i_pair = AI(GRX_RPI * npair); out->rec.i_chain = AI(GRX_RCI * nbody); out->rec.i_weld = AI(GRX_RWI * nweld);
    out->rec.i_act = AI(GRX_RAI * nu); out->rec.i_dof = AI(GRX_RDI * nv); out->rec.i_mpair = AI(GRX_RMI * nmp);
    out->rec.f_body = AF(GRX_RBF * nbody); out->rec.f_jnt = AF(GRX_RJF * njnt); out->rec.f_pair = AF(GRX_RPF * npair); out->rec.f_weld = AF(GRX_RWF * nweld); out->rec.f_eq = AF(GRX_REF * neq);
    out->rec.f_act = AF(GRX_RAF * nu); out->rec.f_dof = AF(GRX_RDF * nv); out->rec.f_ten = AF(GRX_RTF * nten); out->rec.f_mpair = AF(nmp);
    out->rec_i1 = (int)out->i.size(); out->rec_f1 = (int)out->f.size();
  }
  const GrxModel h = grx_bind_model(*out, out->f.data(), out->i.data());
  int32_t* ri = out->i.data(); float* rf = out->f.data();
  auto prm = [&](float* P, const float* solref, const float* solimp, float margin, float dA, float aux) {
    P[GRX_PRM_SOLREF] = solref[0]; P[GRX_PRM_SOLREF + 1] = solref[1];
    for (int k = 0; k < 5; k++) P[GRX_PRM_SOLIMP + k] = solimp[k];
    P[GRX_PRM_MARGIN] = margin; P[GRX_PRM_DA] = dA; P[GRX_PRM_AUX] = aux;
  };
  const bool has_bshift = C("body_shift") >= nbody;
  for (int b = 0; b < nbody; b++) {
    int32_t* I = ri + out->rec.i_body + GRX_RBI * b; float* F = rf + out->rec.f_body + GRX_RBF * b;
    const int mid = h.body_mocapid[b], ja = h.body_jntadr[b], jn = h.body_jntnum[b], jf = (jn >= 1 && ja >= 0 && ja < njnt) ? ja : -1;
    I[0] = mid; I[1] = ja; I[2] = jn; I[3] = (jn == 1 && jf >= 0) ? h.jnt_type[jf] : -1; I[4] = jf >= 0 ? h.jnt_qposadr[jf] : 0;
    I[5] = ((has_bshift && h.body_shift[b]) ? 1 : 0) | ((mid >= 0 || (jn == 1 && jf >= 0 && h.jnt_type[jf] == 0)) ? 2 : 0);
    I[6] = jf >= 0 ? h.jnt_type[jf] : -1; I[7] = 0;
    for (int s_ = 0; s_ < 8; s_++) I[8 + s_] = s_ < njump ? h.body_jump[s_ * nbody + b] : 0;
    for (int k = 0; k < 3; k++) F[k] = h.body_pos[3 * b + k];
    F[3] = jf >= 0 ? h.qpos0[h.jnt_qposadr[jf]] : 0.0f;
    for (int k = 0; k < 4; k++) F[4 + k] = h.body_quat[4 * b + k];
    for (int k = 0; k < 3; k++) { F[8 + k] = jf >= 0 ? h.jnt_pos[3 * jf + k] : 0.0f; F[12 + k] = jf >= 0 ? h.jnt_axis[3 * jf + k] : 0.0f; }
    int32_t* Cc = ri + out->rec.i_chain + GRX_RCI * b;
    Cc[0] = h.dof_chainmask[2 * b]; Cc[1] = h.dof_chainmask[2 * b + 1]; Cc[2] = h.body_rootid[b]; Cc[3] = 0;
  }
  for (int j = 0; j < njnt; j++) {
    int32_t* I = ri + out->rec.i_jnt + GRX_RJI * j; float* F = rf + out->rec.f_jnt + GRX_RJF * j;
    const int b = h.jnt_bodyid[j], da = h.jnt_dofadr[j], qa = h.jnt_qposadr[j], jt = h.jnt_type[j];
    I[0] = qa; I[1] = jt; I[2] = b; I[3] = da; I[4] = h.body_parent[b]; I[5] = h.body_rootid[b]; I[6] = (h.jnt_limited[j] && jt >= 2) ? 1 : 0; I[7] = 0;
    for (int k = 0; k < 3; k++) { F[k] = h.jnt_pos[3 * j + k]; F[4 + k] = h.jnt_axis[3 * j + k]; }
    F[3] = h.qpos0[qa]; F[8] = h.jnt_range[2 * j]; F[9] = h.jnt_range[2 * j + 1];
    prm(F + 12, h.jnt_solref + 2 * j, h.jnt_solimp + 5 * j, h.jnt_margin[j], h.dof_invweight0[da], 0.0f);
  }
  for (int p = 0; p < npair; p++) {
    int32_t* I = ri + out->rec.i_pair + GRX_RPI * p; float* F = rf + out->rec.f_pair + GRX_RPF * p;
    const int g1 = h.pair_geom1[p], g2 = h.pair_geom2[p];
    I[0] = h.pair_condim[p]; I[1] = g1; I[2] = g2; I[3] = h.geom_bodyid[g1]; I[4] = h.geom_bodyid[g2]; I[5] = h.pair_span[p]; I[6] = I[7] = 0;
    F[0] = h.pair_margin[p]; F[1] = h.pair_gap[p]; F[2] = h.pair_margin[p] - h.pair_gap[p];
    for (int k = 0; k < 5; k++) F[3 + k] = h.pair_friction[5 * p + k];
    const float tran = h.geom_invweight0[2 * g1] + h.geom_invweight0[2 * g2];
    prm(F + 8, h.pair_solref + 2 * p, h.pair_solimp + 5 * p, F[2], tran, h.pair_friction[5 * p]);
    F[8 + GRX_PRM_DA2] = (float)h.pair_condim[p];   // the row-parameter pass tells condim 1 from the pyramidal ones here
    for (int k = 0; k < 3; k++) { F[20 + k] = h.geom_size[3 * g1 + k]; F[24 + k] = h.geom_size[3 * g2 + k]; }
  }
  for (int w = 0; w < nweld; w++) {
    int32_t* I = ri + out->rec.i_weld + GRX_RWI * w; float* F = rf + out->rec.f_weld + GRX_RWF * w;
    const int e = h.weld_eq[w], b0 = h.eq_obj1[e], b1 = h.eq_obj2[e];
    I[0] = e; I[1] = b0; I[2] = b1; I[3] = h.weld_row[w];
    I[4] = h.dof_chainmask[2 * b0]; I[5] = h.dof_chainmask[2 * b0 + 1]; I[6] = h.body_rootid[b0];
    I[7] = h.dof_chainmask[2 * b1]; I[8] = h.dof_chainmask[2 * b1 + 1]; I[9] = h.body_rootid[b1]; I[10] = I[11] = 0;
    for (int k = 0; k < 11; k++) F[k] = h.eq_data[11 * e + k];
    for (int k = 0; k < 14; k++) F[12 + k] = h.eq_relpose[14 * e + k];
  }
  for (int e = 0; e < neq; e++) {
    float* F = rf + out->rec.f_eq + GRX_REF * e;
    prm(F, h.eq_solref + 2 * e, h.eq_solimp + 5 * e, 0.0f, h.eq_invweight[2 * e], 0.0f);
    F[GRX_PRM_DA2] = h.eq_invweight[2 * e + 1];
  }
  for (int a = 0; a < nu; a++) {
    int32_t* I = ri + out->rec.i_act + GRX_RAI * a; float* F = rf + out->rec.f_act + GRX_RAF * a;
    const int j = h.act_trnid[a];
    I[0] = h.jnt_qposadr[j]; I[1] = h.jnt_dofadr[j]; I[2] = h.act_ctrllimited[a]; I[3] = h.act_gaintype[a]; I[4] = h.act_biastype[a]; I[5] = h.act_forcelimited[a]; I[6] = I[7] = 0;
    F[0] = h.act_gear[a]; F[1] = h.act_ctrlrange[2 * a]; F[2] = h.act_ctrlrange[2 * a + 1];
    for (int k = 0; k < 3; k++) { F[3 + k] = h.act_gainprm[3 * a + k]; F[6 + k] = h.act_biasprm[3 * a + k]; }
    F[9] = h.act_forcerange[2 * a]; F[10] = h.act_forcerange[2 * a + 1];
  }
  const bool has_dofprm = C("dof_solref") >= 2 * nv && C("dof_solimp") >= 5 * nv;
  for (int q = 0; q < nv; q++) {
    int32_t* I = ri + out->rec.i_dof + GRX_RDI * q; float* F = rf + out->rec.f_dof + GRX_RDF * q;
    const int j = h.dof_jntid[q], e0 = h.dof_cvelstart[q], bb = h.dof_bodyid[e0 >= 0 ? e0 : 0];
    I[0] = h.jnt_qposadr[j]; I[1] = h.jnt_type[j]; I[2] = h.jnt_dofadr[j]; I[3] = e0; I[4] = bb; I[5] = h.body_dofadr[bb] + h.body_dofnum[bb] - 1; I[6] = h.dof_bodyid[q]; I[7] = 0;
    F[0] = h.dof_damping[q]; F[1] = h.jnt_stiffness[j]; F[2] = h.jnt_springref[j]; F[3] = h.dof_armature[q];
    if (has_dofprm) prm(F + 4, h.dof_solref + 2 * q, h.dof_solimp + 5 * q, 0.0f, h.dof_invweight0[q], h.dof_frictionloss[q]);
  }
  for (int e = 0; e < nmp; e++) {
    int32_t* I = ri + out->rec.i_mpair + GRX_RMI * e;
    I[0] = h.mpair_i[e]; I[1] = h.mpair_j[e]; I[2] = h.dof_bodyid[h.mpair_i[e]]; I[3] = 0;
    rf[out->rec.f_mpair + e] = h.dof_armature[h.mpair_i[e]];
  }
  for (int t = 0; t < nten; t++) {
    float* F = rf + out->rec.f_ten + GRX_RTF * t;
    prm(F, h.tendon_solref + 2 * t, h.tendon_solimp + 5 * t, h.tendon_margin[t], h.tendon_invweight0[t], 0.0f);
  }
}
