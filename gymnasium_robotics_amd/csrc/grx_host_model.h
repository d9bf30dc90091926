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
  GrxModel proto;             // scalar members filled; pointers unset
};

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
  return m;
}

inline int grx_find_table(const GrxPackedModel& p, const char* name) {
  for (size_t k = 0; k < p.name.size(); k++) if (p.name[k] == name) return (int)k;
  return -1;
}
