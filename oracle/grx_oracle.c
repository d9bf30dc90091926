/* grx_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Single-world, fp64, deliberately plain restatement of the arithmetic the
 * reference delegates to the third-party MuJoCo library at
 *   /root/reference/gymnasium_robotics/envs/robot_env.py:341  (mujoco.mj_step(model, data, nstep))
 *   /root/reference/gymnasium_robotics/envs/robot_env.py:315  (mujoco.mj_forward)
 *   /root/reference/gymnasium_robotics/envs/robot_env.py:307  (mujoco.mj_resetData)
 *   /root/reference/gymnasium_robotics/utils/mujoco_utils.py:115,125 (mujoco.mj_jacSite)
 * MuJoCo itself (pyproject.toml:27 "mujoco>=2.2.0", unpinned) is NOT present in
 * /root/reference nor installable here, so this file restates MuJoCo's published
 * computation pipeline:
 *   kinematics -> com/cdof -> CRBA -> L'DL -> collision -> constraint rows
 *   -> velocity stage (cvel, passive, aref, RNE bias) -> actuation
 *   -> qacc_smooth -> constraint solve (exact Newton on the primal problem)
 *   -> semi-implicit Euler with implicit joint damping.
 *
 * PARITY UNPINNED for this file (the physics): the reference's tests hold no
 * post-mj_step golden vectors; it is checked only by the
 * known-answer anchors in tests/ (documented Fetch start pose, rest heights,
 * conservation laws, analytic contact cases).  The Python task layers on top of
 * it (oracle/*_oracle.py) ARE pinned: the reference's own step() / reset() code,
 * executed on this physics through tests/ref_harness.py, reproduces them bit
 * for bit (tests/test_cpu_reference_task_layer.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The model tables come from include/grx_model_fields.def.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/grx_model.h"

#define MINVAL 1e-15
#define MAXCON 128
#define MAXEFC 768
#define MINIMP 0.0001
#define MAXIMP 0.9999

enum { EFC_EQUALITY = 0, EFC_FRICTION = 1, EFC_LIMIT = 2, EFC_CONTACT = 3, EFC_TLIMIT = 4 };

typedef struct {
  double dist, pos[3], frame[9];
  int pair, geom1, geom2, dim, efc_address;
  double includemargin, friction[5], solref[2], solimp[5];
} orc_contact;

typedef struct orc_sim {
  /* owned model blob */
  int32_t *H, *I;
  double* F;
  grx_model_view m;
  int nq, nv, nu, nbody, njnt, ngeom, nsite, nmocap, neq, npair;
  /* state */
  double time, *qpos, *qvel, *ctrl, *mocap_pos, *mocap_quat, *qacc_warmstart;
  double shift[7]; /* per-world offset t[3] and rotation q[4] (flag 2 members: x <- R(q) x + t; adroit_pen.py:381 model.body_quat edit) of the model's shift group (body_shift / geom_shift / site_shift): model.body_pos edits of the reference (adroit_hammer.py:374-376) */
  int noslip_iter_done;
  /* position stage */
  double *xpos, *xquat, *xmat, *xipos, *xanchor, *xaxis, *geom_xpos, *geom_xmat, *site_xpos, *site_xmat;
  double *subtree_com, *subtree_mass, *cinert /*36/body*/, *crb /*36/body*/, *cdof /*6/dof*/, *cdof_dot;
  double *M, *L /*dense chol of M (reverse order)*/, *cvel, *cacc, *cfrc;
  double *qfrc_bias, *qfrc_passive, *qfrc_actuator, *qfrc_smooth, *qacc_smooth, *qfrc_constraint, *qacc, *act_force;
  double* touch; /* one value per touch sensor of the touch_* tables */
  /* constraints */
  int ncon, nefc, ne, nf, nl, ntl /* tendon-limit rows among the nl limit rows */;
  orc_contact con[MAXCON];
  double *efc_J /*MAXEFC*nv*/, efc_pos[MAXEFC], efc_margin[MAXEFC], efc_D[MAXEFC], efc_R[MAXEFC], efc_aref[MAXEFC],
      efc_vel[MAXEFC], efc_force[MAXEFC], efc_frictionloss[MAXEFC], efc_diagApprox[MAXEFC], efc_KBIP[4 * MAXEFC];
  int efc_type[MAXEFC], efc_id[MAXEFC];
  /* diagnostics */
  int solver_iter, bad_state, unsupported_hits;
  long mesh_candidates, mesh_contacts; /* hull pairs that reached the convex routine / contacts it produced (cumulative) */
  double solver_gradnorm;
  double min_activation_gap; /* min over steps/rows of |dist - margin|: how close any unilateral row came to (de)activating */
  int opt_disable_mesh_plane; /* test knob */
} orc_sim;

/* ------------------------------------------------------------------ small math */
static void cross3(double* r, const double* a, const double* b) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static double norm3(const double* a) { return sqrt(dot3(a, a)); }
static void mulMatVec3(double* r, const double* m, const double* v) {
  double x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2], y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2],
         z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void mulMatTVec3(double* r, const double* m, const double* v) {
  double x = m[0] * v[0] + m[3] * v[1] + m[6] * v[2], y = m[1] * v[0] + m[4] * v[1] + m[7] * v[2],
         z = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void mulQuat(double* r, const double* a, const double* b) {
  double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
static void normalize4(double* q) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
static void quat2mat(double* m, const double* q) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}
static void axisAngle2Quat(double* q, const double* axis, double angle) {
  double s = sin(0.5 * angle);
  q[0] = cos(0.5 * angle); q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
static void mulMat3(double* r, const double* a, const double* b) {
  double t[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) t[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
  memcpy(r, t, sizeof(t));
}

/* ------------------------------------------------------------------ create / destroy */
#define ALLOC(n) ((double*)calloc((size_t)((n) > 0 ? (n) : 1), sizeof(double)))

orc_sim* orc_create(const int32_t* H, int nH, const int32_t* I, int nI, const double* F, int nF) {
  orc_sim* s = (orc_sim*)calloc(1, sizeof(orc_sim));
  s->shift[3] = 1.0;   /* identity rotation of the shift group */
  s->H = (int32_t*)malloc(sizeof(int32_t) * (size_t)nH); memcpy(s->H, H, sizeof(int32_t) * (size_t)nH);
  s->I = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nI + 1)); memcpy(s->I, I, sizeof(int32_t) * (size_t)nI);
  s->F = (double*)malloc(sizeof(double) * (size_t)(nF + 1)); memcpy(s->F, F, sizeof(double) * (size_t)nF);
  grx_model_view_init(&s->m, s->H, s->I, s->F);
  const int32_t* d = s->m.dims;
  s->nq = d[GRX_NQ]; s->nv = d[GRX_NV]; s->nu = d[GRX_NU]; s->nbody = d[GRX_NBODY]; s->njnt = d[GRX_NJNT];
  s->ngeom = d[GRX_NGEOM]; s->nsite = d[GRX_NSITE]; s->nmocap = d[GRX_NMOCAP]; s->neq = d[GRX_NEQ]; s->npair = d[GRX_NPAIR];
  int nq = s->nq, nv = s->nv, nb = s->nbody;
  s->qpos = ALLOC(nq); s->qvel = ALLOC(nv); s->ctrl = ALLOC(s->nu); s->mocap_pos = ALLOC(3 * s->nmocap);
  s->mocap_quat = ALLOC(4 * s->nmocap); s->qacc_warmstart = ALLOC(nv);
  s->xpos = ALLOC(3 * nb); s->xquat = ALLOC(4 * nb); s->xmat = ALLOC(9 * nb); s->xipos = ALLOC(3 * nb);
  s->xanchor = ALLOC(3 * s->njnt); s->xaxis = ALLOC(3 * s->njnt);
  s->geom_xpos = ALLOC(3 * s->ngeom); s->geom_xmat = ALLOC(9 * s->ngeom);
  s->site_xpos = ALLOC(3 * s->nsite); s->site_xmat = ALLOC(9 * s->nsite); s->touch = ALLOC(s->m.n_touch_body);
  s->subtree_com = ALLOC(3 * nb); s->subtree_mass = ALLOC(nb); s->cinert = ALLOC(36 * nb); s->crb = ALLOC(36 * nb);
  s->cdof = ALLOC(6 * nv); s->cdof_dot = ALLOC(6 * nv); s->M = ALLOC(nv * nv); s->L = ALLOC(nv * nv);
  s->cvel = ALLOC(6 * nb); s->cacc = ALLOC(6 * nb); s->cfrc = ALLOC(6 * nb);
  s->qfrc_bias = ALLOC(nv); s->qfrc_passive = ALLOC(nv); s->qfrc_actuator = ALLOC(nv); s->qfrc_smooth = ALLOC(nv);
  s->qacc_smooth = ALLOC(nv); s->qfrc_constraint = ALLOC(nv); s->qacc = ALLOC(nv); s->act_force = ALLOC(s->nu);
  s->efc_J = ALLOC(MAXEFC * nv);
  return s;
}

void orc_destroy(orc_sim* s) {
  if (!s) return;
  double* ptrs[] = {s->qpos, s->qvel, s->ctrl, s->mocap_pos, s->mocap_quat, s->qacc_warmstart, s->xpos, s->xquat, s->xmat,
                    s->xipos, s->xanchor, s->xaxis, s->geom_xpos, s->geom_xmat, s->site_xpos, s->site_xmat, s->subtree_com,
                    s->subtree_mass, s->cinert, s->crb, s->cdof, s->cdof_dot, s->M, s->L, s->cvel, s->cacc, s->cfrc,
                    s->qfrc_bias, s->qfrc_passive, s->qfrc_actuator, s->qfrc_smooth, s->qacc_smooth, s->qfrc_constraint,
                    s->qacc, s->act_force, s->efc_J, s->F};
  for (size_t i = 0; i < sizeof(ptrs) / sizeof(ptrs[0]); i++) free(ptrs[i]);
  free(s->H); free(s->I); free(s);
}

/* restates mj_resetData [3P] */
void orc_reset_data(orc_sim* s) {
  memcpy(s->qpos, s->m.qpos0, sizeof(double) * (size_t)s->nq);
  memset(s->qvel, 0, sizeof(double) * (size_t)s->nv);
  memset(s->ctrl, 0, sizeof(double) * (size_t)s->nu);
  memset(s->qacc_warmstart, 0, sizeof(double) * (size_t)s->nv);
  memset(s->qacc, 0, sizeof(double) * (size_t)s->nv);
  memcpy(s->mocap_pos, s->m.mocap_pos0, sizeof(double) * 3 * (size_t)s->nmocap);
  memcpy(s->mocap_quat, s->m.mocap_quat0, sizeof(double) * 4 * (size_t)s->nmocap);
  s->time = 0;
  s->bad_state = 0;
  s->min_activation_gap = 1e30;
}

/* ------------------------------------------------------------------ K1 kinematics */
/* A candidate contact that is rejected because it is (just) outside the margin is as much an activation boundary as a listed contact about to leave it: the fixtures'
 * activation_gap would otherwise miss the contact that an engine with 1e-7 of state rounding lists and this one does not (FetchPush fixture, snapshot 273: 7e-8). */
#define NEAR_MISS(dist_, margin_) do { const double g_ = fabs((dist_) - (margin_)); if (g_ > 0 && g_ < s->min_activation_gap) s->min_activation_gap = g_; } while (0)
static void kinematics(orc_sim* s) {
  const grx_model_view* m = &s->m;
  double* xpos = s->xpos; double* xquat = s->xquat; double* xmat = s->xmat;
  xpos[0] = xpos[1] = xpos[2] = 0; xquat[0] = 1; xquat[1] = xquat[2] = xquat[3] = 0; quat2mat(xmat, xquat);
  /* normalise free-joint quaternions in qpos */
  for (int j = 0; j < s->njnt; j++)
    if (m->jnt_type[j] == GRX_JNT_FREE) normalize4(s->qpos + m->jnt_qposadr[j] + 3);
  for (int i = 1; i < s->nbody; i++) {
    double *p = xpos + 3 * i, *q = xquat + 4 * i;
    int par = m->body_parent[i];
    if (m->body_mocapid[i] >= 0) {
      int id = m->body_mocapid[i];
      memcpy(p, s->mocap_pos + 3 * id, 3 * sizeof(double));
      memcpy(q, s->mocap_quat + 4 * id, 4 * sizeof(double));
      normalize4(q);
    } else {
      int jn = m->body_jntnum[i], ja = m->body_jntadr[i];
      if (jn == 1 && m->jnt_type[ja] == GRX_JNT_FREE) {
        const double* qp = s->qpos + m->jnt_qposadr[ja];
        memcpy(p, qp, 3 * sizeof(double)); memcpy(q, qp + 3, 4 * sizeof(double));
        memcpy(s->xanchor + 3 * ja, p, 3 * sizeof(double));
        s->xaxis[3 * ja] = 0; s->xaxis[3 * ja + 1] = 0; s->xaxis[3 * ja + 2] = 1;
      } else {
        double v[3];
        mulMatVec3(v, xmat + 9 * par, m->body_pos + 3 * i);
        p[0] = xpos[3 * par] + v[0]; p[1] = xpos[3 * par + 1] + v[1]; p[2] = xpos[3 * par + 2] + v[2];
        if (m->body_shift[i]) { p[0] += s->shift[0]; p[1] += s->shift[1]; p[2] += s->shift[2]; }   /* child of the (world-fixed) shift group */
        mulQuat(q, xquat + 4 * par, m->body_quat + 4 * i);
        for (int k = 0; k < jn; k++) {
          int j = ja + k;
          double R[9]; quat2mat(R, q);
          double* anchor = s->xanchor + 3 * j; double* axis = s->xaxis + 3 * j;
          mulMatVec3(anchor, R, m->jnt_pos + 3 * j);
          anchor[0] += p[0]; anchor[1] += p[1]; anchor[2] += p[2];
          mulMatVec3(axis, R, m->jnt_axis + 3 * j);
          double dq = s->qpos[m->jnt_qposadr[j]] - m->qpos0[m->jnt_qposadr[j]];
          if (m->jnt_type[j] == GRX_JNT_SLIDE) {
            p[0] += axis[0] * dq; p[1] += axis[1] * dq; p[2] += axis[2] * dq;
          } else if (m->jnt_type[j] == GRX_JNT_HINGE) {
            double qr[4], qn[4];
            axisAngle2Quat(qr, m->jnt_axis + 3 * j, dq);
            mulQuat(qn, q, qr); memcpy(q, qn, sizeof(qn));
            /* keep the anchor fixed: xpos = anchor - R_new * jnt_pos */
            double Rn[9], off[3]; quat2mat(Rn, q); mulMatVec3(off, Rn, m->jnt_pos + 3 * j);
            p[0] = anchor[0] - off[0]; p[1] = anchor[1] - off[1]; p[2] = anchor[2] - off[2];
          }
        }
      }
    }
    normalize4(q);
    quat2mat(xmat + 9 * i, q);
    double v[3]; mulMatVec3(v, xmat + 9 * i, m->body_ipos + 3 * i);
    s->xipos[3 * i] = p[0] + v[0]; s->xipos[3 * i + 1] = p[1] + v[1]; s->xipos[3 * i + 2] = p[2] + v[2];
  }
  for (int g = 0; g < s->ngeom; g++) {
    int b = m->geom_bodyid[g]; double v[3], R[9];
    mulMatVec3(v, xmat + 9 * b, m->geom_pos + 3 * g);
    for (int k = 0; k < 3; k++) v[k] += xpos[3 * b + k];
    quat2mat(R, m->geom_quat + 4 * g); mulMat3(s->geom_xmat + 9 * g, xmat + 9 * b, R);
    if (m->geom_shift[g] == 2) { double Rg[9], t[9], w[3]; quat2mat(Rg, s->shift + 3); mulMatVec3(w, Rg, v); mulMat3(t, Rg, s->geom_xmat + 9 * g); memcpy(v, w, sizeof w); memcpy(s->geom_xmat + 9 * g, t, sizeof t); }
    for (int k = 0; k < 3; k++) s->geom_xpos[3 * g + k] = v[k] + (m->geom_shift[g] ? s->shift[k] : 0.0);
  }
  for (int g = 0; g < s->nsite; g++) {
    int b = m->site_bodyid[g]; double v[3], R[9];
    mulMatVec3(v, xmat + 9 * b, m->site_pos + 3 * g);
    for (int k = 0; k < 3; k++) v[k] += xpos[3 * b + k];
    quat2mat(R, m->site_quat + 4 * g); mulMat3(s->site_xmat + 9 * g, xmat + 9 * b, R);
    if (m->site_shift[g] == 2) { double Rg[9], t[9], w[3]; quat2mat(Rg, s->shift + 3); mulMatVec3(w, Rg, v); mulMat3(t, Rg, s->site_xmat + 9 * g); memcpy(v, w, sizeof w); memcpy(s->site_xmat + 9 * g, t, sizeof t); }
    for (int k = 0; k < 3; k++) s->site_xpos[3 * g + k] = v[k] + (m->site_shift[g] ? s->shift[k] : 0.0);
  }
}

/* ------------------------------------------------------------------ K2 com, cinert, cdof */
/* 6x6 spatial inertia of a rigid body about reference point c, [rot; lin] ordering:
 * kinetic energy = 1/2 v' I v with v = [omega; velocity of the point of the body at c] */
static void spatial_inertia(double* I6, double mass, const double* Ic /*3x3 world*/, const double* r /*com - c*/) {
  double rx[9] = {0, -r[2], r[1], r[2], 0, -r[0], -r[1], r[0], 0};
  memset(I6, 0, 36 * sizeof(double));
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double rr = 0;
      for (int k = 0; k < 3; k++) rr += rx[3 * k + i] * rx[3 * k + j]; /* rx' rx */
      I6[6 * i + j] = Ic[3 * i + j] + mass * rr;
      I6[6 * i + 3 + j] = mass * rx[3 * i + j];
      I6[6 * (3 + i) + j] = mass * rx[3 * j + i];
    }
  for (int i = 0; i < 3; i++) I6[6 * (3 + i) + 3 + i] = mass;
}

static void com_pos(orc_sim* s) {
  const grx_model_view* m = &s->m;
  int nb = s->nbody;
  for (int i = 0; i < nb; i++) {
    s->subtree_mass[i] = m->body_mass[i];
    for (int k = 0; k < 3; k++) s->subtree_com[3 * i + k] = m->body_mass[i] * s->xipos[3 * i + k];
  }
  for (int i = nb - 1; i > 0; i--) {
    int p = m->body_parent[i];
    s->subtree_mass[p] += s->subtree_mass[i];
    for (int k = 0; k < 3; k++) s->subtree_com[3 * p + k] += s->subtree_com[3 * i + k];
  }
  for (int i = 0; i < nb; i++)
    for (int k = 0; k < 3; k++)
      s->subtree_com[3 * i + k] = s->subtree_mass[i] > MINVAL ? s->subtree_com[3 * i + k] / s->subtree_mass[i] : s->xipos[3 * i + k];
  for (int i = 1; i < nb; i++) {
    const double* c = s->subtree_com + 3 * m->body_rootid[i];
    const double* R = s->xmat + 9 * i; const double* in = m->body_inertia + 6 * i;
    double Ib[9] = {in[0], in[3], in[4], in[3], in[1], in[5], in[4], in[5], in[2]}, t[9], Rt[9], Iw[9];
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) Rt[3 * a + b] = R[3 * b + a];
    mulMat3(t, R, Ib); mulMat3(Iw, t, Rt);
    double r[3] = {s->xipos[3 * i] - c[0], s->xipos[3 * i + 1] - c[1], s->xipos[3 * i + 2] - c[2]};
    spatial_inertia(s->cinert + 36 * i, m->body_mass[i], Iw, r);
  }
  for (int j = 0; j < s->njnt; j++) {
    int b = m->jnt_bodyid[j], da = m->jnt_dofadr[j];
    const double* c = s->subtree_com + 3 * m->body_rootid[b];
    double off[3] = {c[0] - s->xanchor[3 * j], c[1] - s->xanchor[3 * j + 1], c[2] - s->xanchor[3 * j + 2]};
    if (m->jnt_type[j] == GRX_JNT_SLIDE) {
      double* cd = s->cdof + 6 * da;
      cd[0] = cd[1] = cd[2] = 0; memcpy(cd + 3, s->xaxis + 3 * j, 3 * sizeof(double));
    } else if (m->jnt_type[j] == GRX_JNT_HINGE) {
      double* cd = s->cdof + 6 * da;
      memcpy(cd, s->xaxis + 3 * j, 3 * sizeof(double)); cross3(cd + 3, s->xaxis + 3 * j, off);
    } else if (m->jnt_type[j] == GRX_JNT_FREE) {
      for (int k = 0; k < 3; k++) {
        double* cd = s->cdof + 6 * (da + k);
        memset(cd, 0, 6 * sizeof(double)); cd[3 + k] = 1;
      }
      for (int k = 0; k < 3; k++) {
        double* cd = s->cdof + 6 * (da + 3 + k);
        double ax[3] = {s->xmat[9 * b + k], s->xmat[9 * b + 3 + k], s->xmat[9 * b + 6 + k]};
        memcpy(cd, ax, sizeof(ax)); cross3(cd + 3, ax, off);
      }
    }
  }
}

/* ------------------------------------------------------------------ K3/K4 CRBA + factor */
static void crb_and_factor(orc_sim* s) {
  const grx_model_view* m = &s->m;
  int nv = s->nv, nb = s->nbody;
  memcpy(s->crb, s->cinert, sizeof(double) * 36 * (size_t)nb);
  for (int i = nb - 1; i > 0; i--) {
    int p = m->body_parent[i];
    if (p > 0) for (int k = 0; k < 36; k++) s->crb[36 * p + k] += s->crb[36 * i + k];
  }
  memset(s->M, 0, sizeof(double) * (size_t)(nv * nv));
  for (int i = 0; i < nv; i++) {
    double buf[6]; const double* I6 = s->crb + 36 * m->dof_bodyid[i]; const double* cd = s->cdof + 6 * i;
    for (int a = 0; a < 6; a++) { buf[a] = 0; for (int b = 0; b < 6; b++) buf[a] += I6[6 * a + b] * cd[b]; }
    for (int j = i; j >= 0; j = m->dof_parentid[j]) {
      double v = 0; for (int a = 0; a < 6; a++) v += s->cdof[6 * j + a] * buf[a];
      s->M[i * nv + j] = s->M[j * nv + i] = v;
    }
    s->M[i * nv + i] += m->dof_armature[i];
  }
}

/* dense factorisation A = U' U computed from the LAST dof towards the first
 * (same elimination order as MuJoCo's L'DL: leaves before root), in place on a copy. */
static int chol_reverse(double* U, const double* A, int n) {
  /* U upper-left... we compute A = L' L with L lower triangular: eliminate from i = n-1 down */
  memcpy(U, A, sizeof(double) * (size_t)(n * n));
  for (int i = n - 1; i >= 0; i--) {
    double d = U[i * n + i];
    if (d < MINVAL) return -1;
    d = sqrt(d); U[i * n + i] = d;
    for (int j = 0; j < i; j++) U[i * n + j] /= d;
    for (int j = 0; j < i; j++)
      for (int k = 0; k <= j; k++) U[j * n + k] -= U[i * n + j] * U[i * n + k];
  }
  return 0;
}
/* solve A x = b given L from chol_reverse (A = L' L, L lower) */
static void chol_reverse_solve(const double* L, int n, double* x) {
  /* L' y = b : L' is upper; solve from last to first */
  for (int i = n - 1; i >= 0; i--) {
    double v = x[i];
    for (int k = i + 1; k < n; k++) v -= L[k * n + i] * x[k];
    x[i] = v / L[i * n + i];
  }
  /* L x = y : forward */
  for (int i = 0; i < n; i++) {
    double v = x[i];
    for (int k = 0; k < i; k++) v -= L[i * n + k] * x[k];
    x[i] = v / L[i * n + i];
  }
}

/* ------------------------------------------------------------------ Jacobians */
/* translational (jacp 3xnv) and rotational (jacr 3xnv) Jacobian of world point fixed to body */
static void jac_point(const orc_sim* s, int body, const double* point, double* jacp, double* jacr) {
  const grx_model_view* m = &s->m; int nv = s->nv;
  if (jacp) memset(jacp, 0, sizeof(double) * 3 * (size_t)nv);
  if (jacr) memset(jacr, 0, sizeof(double) * 3 * (size_t)nv);
  if (body <= 0 || m->body_dofnum[body] == 0) {
    /* climb to a body with dofs */
    while (body > 0 && m->body_dofnum[body] == 0) body = m->body_parent[body];
    if (body <= 0) return;
  }
  const double* c = s->subtree_com + 3 * m->body_rootid[body];
  double off[3] = {point[0] - c[0], point[1] - c[1], point[2] - c[2]};
  int d = m->body_dofadr[body] + m->body_dofnum[body] - 1;
  for (; d >= 0; d = m->dof_parentid[d]) {
    const double* cd = s->cdof + 6 * d;
    if (jacr) { jacr[d] = cd[0]; jacr[nv + d] = cd[1]; jacr[2 * nv + d] = cd[2]; }
    if (jacp) {
      double t[3]; cross3(t, cd, off);
      jacp[d] = cd[3] + t[0]; jacp[nv + d] = cd[4] + t[1]; jacp[2 * nv + d] = cd[5] + t[2];
    }
  }
}

/* restates mujoco.mj_jacSite [3P] as used by mujoco_utils.py:110-127 */
void orc_jac_site(orc_sim* s, int site, double* jacp, double* jacr) {
  jac_point(s, s->m.site_bodyid[site], s->site_xpos + 3 * site, jacp, jacr);
}

/* ------------------------------------------------------------------ K8 collision */
static void make_frame(double* f) {
  /* f[0:3] = normal (unit). pick y by MuJoCo's rule, orthogonalise, z = x cross y */
  double* x = f; double* y = f + 3; double* z = f + 6;
  if (x[1] < 0.5 && x[1] > -0.5) { y[0] = 0; y[1] = 1; y[2] = 0; } else { y[0] = 0; y[1] = 0; y[2] = 1; }
  double d = dot3(x, y); y[0] -= d * x[0]; y[1] -= d * x[1]; y[2] -= d * x[2];
  double n = norm3(y); y[0] /= n; y[1] /= n; y[2] /= n;
  cross3(z, x, y);
}

static orc_contact* add_contact(orc_sim* s, int pair, const double* pos, const double* normal, double dist) {
  const grx_model_view* m = &s->m;
  if (s->ncon >= MAXCON) return NULL;
  orc_contact* c = &s->con[s->ncon++];
  c->pair = pair; c->geom1 = m->pair_geom1[pair]; c->geom2 = m->pair_geom2[pair];
  c->dist = dist; memcpy(c->pos, pos, 3 * sizeof(double)); memcpy(c->frame, normal, 3 * sizeof(double));
  make_frame(c->frame);
  c->dim = m->pair_condim[pair];
  c->includemargin = m->pair_margin[pair] - m->pair_gap[pair];
  memcpy(c->friction, m->pair_friction + 5 * pair, 5 * sizeof(double));
  memcpy(c->solref, m->pair_solref + 2 * pair, 2 * sizeof(double));
  memcpy(c->solimp, m->pair_solimp + 5 * pair, 5 * sizeof(double));
  c->efc_address = -1;
  return c;
}

static void collide_plane_box(orc_sim* s, int pair, int g1, int g2, double margin) {
  const grx_model_view* m = &s->m;
  const double* pp = s->geom_xpos + 3 * g1; const double* pm = s->geom_xmat + 9 * g1;
  const double* bp = s->geom_xpos + 3 * g2; const double* bm = s->geom_xmat + 9 * g2; const double* sz = m->geom_size + 3 * g2;
  double n[3] = {pm[2], pm[5], pm[8]};
  int cnt = 0;
  for (int c = 0; c < 8 && cnt < 4; c++) {
    double loc[3] = {(c & 1 ? sz[0] : -sz[0]), (c & 2 ? sz[1] : -sz[1]), (c & 4 ? sz[2] : -sz[2])}, w[3];
    mulMatVec3(w, bm, loc); w[0] += bp[0]; w[1] += bp[1]; w[2] += bp[2];
    double d[3] = {w[0] - pp[0], w[1] - pp[1], w[2] - pp[2]};
    double dist = dot3(d, n);
    NEAR_MISS(dist, margin);
    if (dist > margin) continue;
    double pos[3] = {w[0] - 0.5 * dist * n[0], w[1] - 0.5 * dist * n[1], w[2] - 0.5 * dist * n[2]};
    add_contact(s, pair, pos, n, dist); cnt++;
  }
}

static void collide_plane_sphere(orc_sim* s, int pair, int g1, int g2, double margin) {
  const grx_model_view* m = &s->m;
  const double* pm = s->geom_xmat + 9 * g1; double n[3] = {pm[2], pm[5], pm[8]};
  const double* c = s->geom_xpos + 3 * g2; double r = m->geom_size[3 * g2];
  double d[3] = {c[0] - s->geom_xpos[3 * g1], c[1] - s->geom_xpos[3 * g1 + 1], c[2] - s->geom_xpos[3 * g1 + 2]};
  double dist = dot3(d, n) - r;
  NEAR_MISS(dist, margin);
  if (dist > margin) return;
  double pos[3] = {c[0] - n[0] * (r + 0.5 * dist), c[1] - n[1] * (r + 0.5 * dist), c[2] - n[2] * (r + 0.5 * dist)};
  add_contact(s, pair, pos, n, dist);
}

/* sphere (geom1) vs box (geom2): closest point of the box to the sphere centre; normal from the sphere to the box */
static void collide_sphere_box(orc_sim* s, int pair, int g1, int g2, double margin) {
  const grx_model_view* m = &s->m;
  const double* c = s->geom_xpos + 3 * g1; double r = m->geom_size[3 * g1];
  const double* bp = s->geom_xpos + 3 * g2; const double* bm = s->geom_xmat + 9 * g2; const double* sz = m->geom_size + 3 * g2;
  double dw[3] = {c[0] - bp[0], c[1] - bp[1], c[2] - bp[2]}, loc[3], cl[3];
  mulMatTVec3(loc, bm, dw);
  int inside = 1;
  for (int k = 0; k < 3; k++) { cl[k] = fmin(sz[k], fmax(-sz[k], loc[k])); if (cl[k] != loc[k]) inside = 0; }
  double nl[3], dist;
  if (!inside) {
    double dv[3] = {cl[0] - loc[0], cl[1] - loc[1], cl[2] - loc[2]};
    double len = norm3(dv);
    dist = len - r;
    NEAR_MISS(dist, margin);
    if (dist > margin) return;
    for (int k = 0; k < 3; k++) nl[k] = dv[k] / len;
  } else {
    /* centre inside the box: push out through the nearest face */
    int ax = 0; double best = 1e30;
    for (int k = 0; k < 3; k++) { double dd = sz[k] - fabs(loc[k]); if (dd < best) { best = dd; ax = k; } }
    nl[0] = nl[1] = nl[2] = 0; nl[ax] = loc[ax] >= 0 ? -1 : 1;
    dist = -best - r;
  }
  double n[3]; mulMatVec3(n, bm, nl);
  double pos[3] = {c[0] + n[0] * (r + 0.5 * dist), c[1] + n[1] * (r + 0.5 * dist), c[2] + n[2] * (r + 0.5 * dist)};
  add_contact(s, pair, pos, n, dist);
}

/* plane vs capsule: the two end spheres (MuJoCo's plane-capsule routine) */
static void collide_plane_capsule(orc_sim* s, int pair, int g1, int g2, double margin) {
  const grx_model_view* m = &s->m;
  const double* pm = s->geom_xmat + 9 * g1; double n[3] = {pm[2], pm[5], pm[8]};
  const double* c = s->geom_xpos + 3 * g2; const double* R = s->geom_xmat + 9 * g2;
  double r = m->geom_size[3 * g2], hl = m->geom_size[3 * g2 + 1], ax[3] = {R[2], R[5], R[8]};
  for (int e = -1; e <= 1; e += 2) {
    double p[3] = {c[0] + e * hl * ax[0], c[1] + e * hl * ax[1], c[2] + e * hl * ax[2]};
    double d[3] = {p[0] - s->geom_xpos[3 * g1], p[1] - s->geom_xpos[3 * g1 + 1], p[2] - s->geom_xpos[3 * g1 + 2]};
    double dist = dot3(d, n) - r;
    NEAR_MISS(dist, margin);
    if (dist > margin) continue;
    double pos[3] = {p[0] - n[0] * (r + 0.5 * dist), p[1] - n[1] * (r + 0.5 * dist), p[2] - n[2] * (r + 0.5 * dist)};
    add_contact(s, pair, pos, n, dist);
  }
}

/* squared distance from a point (box frame) to the box, and the clamped point */
static double box_point_dist2(const double* sz, const double* p, double* cl) {
  double d2 = 0;
  for (int k = 0; k < 3; k++) { cl[k] = fmin(sz[k], fmax(-sz[k], p[k])); d2 += (p[k] - cl[k]) * (p[k] - cl[k]); }
  return d2;
}
/* sphere of radius r at box-frame point p against the box: contact (normal from the sphere to the box) */
static int sphere_box_local(orc_sim* s, int pair, const double* bp, const double* bm, const double* sz, const double* p, double r, double margin) {
  double cl[3], nl[3], dist;
  double d2 = box_point_dist2(sz, p, cl);
  if (d2 > 0) {
    double len = sqrt(d2); dist = len - r;
    NEAR_MISS(dist, margin);
    if (dist > margin) return 0;
    for (int k = 0; k < 3; k++) nl[k] = (cl[k] - p[k]) / len;
  } else {
    int ax = 0; double best = 1e30;
    for (int k = 0; k < 3; k++) { double dd = sz[k] - fabs(p[k]); if (dd < best) { best = dd; ax = k; } }
    nl[0] = nl[1] = nl[2] = 0; nl[ax] = p[ax] >= 0 ? -1 : 1; dist = -best - r;
  }
  double n[3], pw[3]; mulMatVec3(n, bm, nl); mulMatVec3(pw, bm, p);
  double pos[3] = {pw[0] + bp[0] + n[0] * (r + 0.5 * dist), pw[1] + bp[1] + n[1] * (r + 0.5 * dist), pw[2] + bp[2] + n[2] * (r + 0.5 * dist)};
  add_contact(s, pair, pos, n, dist);
  return 1;
}
/* capsule (geom1) vs box (geom2).  MuJoCo has a dedicated routine whose exact contact placement cannot be checked here;
 * this restatement takes the point of the capsule axis closest to the box (convex 1-D minimisation) as a sphere contact,
 * plus the farther end sphere when it is inside the margin too (capsule lying along a face): at most 2 contacts. */
static void collide_capsule_box(orc_sim* s, int pair, int g1, int g2, double margin) {
  const grx_model_view* m = &s->m;
  const double* c = s->geom_xpos + 3 * g1; const double* R = s->geom_xmat + 9 * g1;
  const double* bp = s->geom_xpos + 3 * g2; const double* bm = s->geom_xmat + 9 * g2; const double* sz = m->geom_size + 3 * g2;
  double r = m->geom_size[3 * g1], hl = m->geom_size[3 * g1 + 1];
  double axw[3] = {R[2], R[5], R[8]}, dw[3] = {c[0] - bp[0], c[1] - bp[1], c[2] - bp[2]}, cen[3], ax[3];
  mulMatTVec3(cen, bm, dw); mulMatTVec3(ax, bm, axw);
  /* golden-section search of t in [-hl, hl] for the axis point closest to the box (distance is convex in t) */
  double lo = -hl, hi = hl, cl[3];
  const double gr = 0.6180339887498949;
  double t1 = hi - gr * (hi - lo), t2 = lo + gr * (hi - lo);
  double p1[3] = {cen[0] + t1 * ax[0], cen[1] + t1 * ax[1], cen[2] + t1 * ax[2]}, p2[3] = {cen[0] + t2 * ax[0], cen[1] + t2 * ax[1], cen[2] + t2 * ax[2]};
  double f1 = box_point_dist2(sz, p1, cl), f2 = box_point_dist2(sz, p2, cl);
  for (int it = 0; it < 48; it++) {
    if (f1 <= f2) { hi = t2; t2 = t1; f2 = f1; t1 = hi - gr * (hi - lo); for (int k = 0; k < 3; k++) p1[k] = cen[k] + t1 * ax[k]; f1 = box_point_dist2(sz, p1, cl); }
    else { lo = t1; t1 = t2; f1 = f2; t2 = lo + gr * (hi - lo); for (int k = 0; k < 3; k++) p2[k] = cen[k] + t2 * ax[k]; f2 = box_point_dist2(sz, p2, cl); }
  }
  double ts = 0.5 * (lo + hi);
  /* snap to an end point when the minimum sits there */
  double pe[3], fe;
  for (int e = -1; e <= 1; e += 2) { for (int k = 0; k < 3; k++) pe[k] = cen[k] + e * hl * ax[k]; fe = box_point_dist2(sz, pe, cl); if (fe <= fmin(f1, f2)) ts = e * hl; }
  /* the axis segment passes through the box (penetration deeper than the radius): every point of the inside stretch has distance 0 and the
   * minimiser above is not unique; take the middle of the stretch (slab clipping of the segment against the three pairs of faces) */
  {
    double ta = -hl, tb = hl; int hit = 1;
    for (int k = 0; k < 3 && hit; k++) {
      if (fabs(ax[k]) < MINVAL) { if (fabs(cen[k]) > sz[k]) hit = 0; }
      else { double u = (-sz[k] - cen[k]) / ax[k], v = (sz[k] - cen[k]) / ax[k]; if (u > v) { double w = u; u = v; v = w; } if (u > ta) ta = u; if (v < tb) tb = v; }
    }
    if (hit && ta < tb) ts = 0.5 * (ta + tb);
  }
  double ps[3] = {cen[0] + ts * ax[0], cen[1] + ts * ax[1], cen[2] + ts * ax[2]};
  if (!sphere_box_local(s, pair, bp, bm, sz, ps, r, margin)) return;
  double te = (ts >= 0) ? -hl : hl;
  if (fabs(te - ts) > 0.2 * hl) {
    double pf[3] = {cen[0] + te * ax[0], cen[1] + te * ax[1], cen[2] + te * ax[2]};
    sphere_box_local(s, pair, bp, bm, sz, pf, r, margin);
  }
}

/* capsule vs capsule: closest points of the two axis segments (clamped), then a sphere-sphere contact between them
 * (normal from geom1 to geom2, position halfway through the overlap).  MuJoCo's routine additionally emits a second
 * contact for exactly parallel axes (|det| < mjMINVAL), a measure-zero configuration that is not restated. */
static void collide_capsule_capsule(orc_sim* s, int pair, int g1, int g2, double margin) {
  const grx_model_view* m = &s->m;
  const double* c1 = s->geom_xpos + 3 * g1; const double* R1 = s->geom_xmat + 9 * g1;
  const double* c2 = s->geom_xpos + 3 * g2; const double* R2 = s->geom_xmat + 9 * g2;
  double r1 = m->geom_size[3 * g1], h1 = m->geom_size[3 * g1 + 1], r2 = m->geom_size[3 * g2], h2 = m->geom_size[3 * g2 + 1];
  double a1[3] = {R1[2], R1[5], R1[8]}, a2[3] = {R2[2], R2[5], R2[8]}, w[3] = {c1[0] - c2[0], c1[1] - c2[1], c1[2] - c2[2]};
  /* minimise |c1 + x1 a1 - c2 - x2 a2|^2 over x1 in [-h1, h1], x2 in [-h2, h2] */
  double b = dot3(a1, a2), d = dot3(a1, w), e = dot3(a2, w), den = 1 - b * b, x1, x2;
  x1 = den > MINVAL ? (b * e - d) / den : 0;
  x1 = fmin(h1, fmax(-h1, x1));
  x2 = b * x1 + e;
  if (x2 > h2) { x2 = h2; x1 = fmin(h1, fmax(-h1, b * x2 - d)); }
  else if (x2 < -h2) { x2 = -h2; x1 = fmin(h1, fmax(-h1, b * x2 - d)); }
  double p1[3], p2[3], n[3];
  for (int k = 0; k < 3; k++) { p1[k] = c1[k] + x1 * a1[k]; p2[k] = c2[k] + x2 * a2[k]; n[k] = p2[k] - p1[k]; }
  double len = norm3(n);
  if (len < MINVAL) { n[0] = 1; n[1] = n[2] = 0; } else { n[0] /= len; n[1] /= len; n[2] /= len; }
  double dist = len - r1 - r2;
  NEAR_MISS(dist, margin);
  if (dist > margin) return;
  double pos[3] = {p1[0] + n[0] * (r1 + 0.5 * dist), p1[1] + n[1] * (r1 + 0.5 * dist), p1[2] + n[2] * (r1 + 0.5 * dist)};
  add_contact(s, pair, pos, n, dist);
}

/* sphere vs sphere, sphere vs capsule (sphere against the closest point of the capsule's axis segment): MuJoCo's analytic routines
 * mjc_SphereSphere / mjc_SphereCapsule [3P] -- normal from geom1 to geom2, position halfway through the overlap */
static void sphere_sphere_raw(orc_sim* s, int pair, const double* c1, double r1, const double* c2, double r2, double margin) {
  double n[3] = {c2[0] - c1[0], c2[1] - c1[1], c2[2] - c1[2]};
  double len = norm3(n);
  double dist = len - r1 - r2;
  NEAR_MISS(dist, margin);
  if (dist > margin) return;
  if (len < MINVAL) { n[0] = 1; n[1] = n[2] = 0; } else { n[0] /= len; n[1] /= len; n[2] /= len; }
  double pos[3] = {c1[0] + n[0] * (r1 + 0.5 * dist), c1[1] + n[1] * (r1 + 0.5 * dist), c1[2] + n[2] * (r1 + 0.5 * dist)};
  add_contact(s, pair, pos, n, dist);
}
static void collide_sphere_sphere(orc_sim* s, int pair, int g1, int g2, double margin) {
  sphere_sphere_raw(s, pair, s->geom_xpos + 3 * g1, s->m.geom_size[3 * g1], s->geom_xpos + 3 * g2, s->m.geom_size[3 * g2], margin);
}
static void collide_sphere_capsule(orc_sim* s, int pair, int g1, int g2, double margin) {
  const grx_model_view* m = &s->m;
  const double* c1 = s->geom_xpos + 3 * g1; const double* c2 = s->geom_xpos + 3 * g2; const double* R2 = s->geom_xmat + 9 * g2;
  double ax[3] = {R2[2], R2[5], R2[8]}, d[3] = {c1[0] - c2[0], c1[1] - c2[1], c1[2] - c2[2]};
  double h = m->geom_size[3 * g2 + 1], x = fmin(h, fmax(-h, dot3(ax, d)));
  double p2[3] = {c2[0] + x * ax[0], c2[1] + x * ax[1], c2[2] + x * ax[2]};
  sphere_sphere_raw(s, pair, c1, m->geom_size[3 * g1], p2, m->geom_size[3 * g2], margin);
}

/* plane vs convex hull of a mesh: deepest hull vertex + up to 3 of its hull neighbours
 * that are also within the margin (restated from memory of MuJoCo's plane-convex routine;
 * unverifiable here -- see DESIGN.md "mesh policy") */
static void collide_plane_mesh(orc_sim* s, int pair, int g1, int g2, double margin) {
  const grx_model_view* m = &s->m;
  if (s->opt_disable_mesh_plane) return;
  const double* pp = s->geom_xpos + 3 * g1; const double* pm = s->geom_xmat + 9 * g1;
  const double* gp = s->geom_xpos + 3 * g2; const double* gm = s->geom_xmat + 9 * g2;
  double n[3] = {pm[2], pm[5], pm[8]}, nl[3];
  mulMatTVec3(nl, gm, n); /* plane normal in mesh frame */
  int adr = m->geom_meshadr[g2], num = m->geom_meshnum[g2];
  double off = dot3(gp, n) - dot3(pp, n);
  int best = -1; double bd = 1e30;
  for (int v = 0; v < num; v++) {
    double d = dot3(m->mesh_vert + 3 * (adr + v), nl) + off;
    if (d < bd) { bd = d; best = v; }
  }
  if (best >= 0) NEAR_MISS(bd, margin);
  if (best < 0 || bd > margin) return;
  int verts[4] = {best, -1, -1, -1}; double dists[4] = {bd, 0, 0, 0}; int cnt = 1;
  int aa = m->mesh_adjadr[adr + best], an = m->mesh_adjnum[adr + best];
  for (int k = 0; k < an && cnt < 4; k++) {
    int v = m->mesh_adj[aa + k];
    double d = dot3(m->mesh_vert + 3 * (adr + v), nl) + off;
    if (d <= margin) { verts[cnt] = v; dists[cnt] = d; cnt++; }
  }
  for (int k = 0; k < cnt; k++) {
    double w[3]; mulMatVec3(w, gm, m->mesh_vert + 3 * (adr + verts[k]));
    double pos[3] = {w[0] + gp[0] - 0.5 * dists[k] * n[0], w[1] + gp[1] - 0.5 * dists[k] * n[1], w[2] + gp[2] - 0.5 * dists[k] * n[2]};
    add_contact(s, pair, pos, n, dists[k]);
  }
}

/* box-box: separating-axis test over the 15 candidate axes, then either
 * face clipping (reference face vs incident face) or edge-edge closest points.
 * Contact convention (MuJoCo): normal points from geom1 to geom2, pos is midway
 * between the two surfaces, dist < 0 when penetrating. */
static int clip_poly(double (*poly)[2], int n, int axis, double lim, double sign) {
  /* keep points with sign*p[axis] <= lim ; Sutherland-Hodgman in 2D */
  double out[16][2]; int no = 0;
  for (int i = 0; i < n; i++) {
    double* a = poly[i]; double* b = poly[(i + 1) % n];
    double da = sign * a[axis] - lim, db = sign * b[axis] - lim;
    if (da <= 0) { out[no][0] = a[0]; out[no][1] = a[1]; no++; }
    if ((da < 0 && db > 0) || (da > 0 && db < 0)) {
      double t = da / (da - db);
      out[no][0] = a[0] + t * (b[0] - a[0]); out[no][1] = a[1] + t * (b[1] - a[1]); no++;
    }
    if (no >= 15) break;
  }
  for (int i = 0; i < no; i++) { poly[i][0] = out[i][0]; poly[i][1] = out[i][1]; }
  return no;
}

static void collide_box_box(orc_sim* s, int pair, int g1, int g2, double margin) {
  const grx_model_view* m = &s->m;
  const double* p1 = s->geom_xpos + 3 * g1; const double* R1 = s->geom_xmat + 9 * g1; const double* a = m->geom_size + 3 * g1;
  const double* p2 = s->geom_xpos + 3 * g2; const double* R2 = s->geom_xmat + 9 * g2; const double* b = m->geom_size + 3 * g2;
  double A[3][3], Bx[3][3]; /* box axes as rows */
  for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) { A[i][k] = R1[3 * k + i]; Bx[i][k] = R2[3 * k + i]; }
  double d[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  double C[3][3], Q[3][3];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { C[i][j] = dot3(A[i], Bx[j]); Q[i][j] = fabs(C[i][j]); }
  double best = -1e30; int code = -1; double bn[3] = {0, 0, 0};
  /* face axes of box1 and box2 */
  for (int i = 0; i < 3; i++) {
    double t = dot3(d, A[i]);
    double sep = fabs(t) - (a[i] + b[0] * Q[i][0] + b[1] * Q[i][1] + b[2] * Q[i][2]);
    NEAR_MISS(sep, margin);
    if (sep > margin) return;
    if (sep > best) { best = sep; code = i; double sg = t < 0 ? -1 : 1; for (int k = 0; k < 3; k++) bn[k] = sg * A[i][k]; }
  }
  for (int j = 0; j < 3; j++) {
    double t = dot3(d, Bx[j]);
    double sep = fabs(t) - (b[j] + a[0] * Q[0][j] + a[1] * Q[1][j] + a[2] * Q[2][j]);
    NEAR_MISS(sep, margin);
    if (sep > margin) return;
    if (sep > best) { best = sep; code = 3 + j; double sg = t < 0 ? -1 : 1; for (int k = 0; k < 3; k++) bn[k] = sg * Bx[j][k]; }
  }
  /* edge-edge axes; prefer faces unless an edge axis is clearly better */
  double ebest = -1e30; int ei = -1, ej = -1; double en[3] = {0, 0, 0};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double ax[3]; cross3(ax, A[i], Bx[j]);
      double l = norm3(ax);
      if (l < 1e-6) continue;
      ax[0] /= l; ax[1] /= l; ax[2] /= l;
      double t = dot3(d, ax);
      double ra = 0, rb = 0;
      for (int k = 0; k < 3; k++) { ra += a[k] * fabs(dot3(A[k], ax)); rb += b[k] * fabs(dot3(Bx[k], ax)); }
      double sep = fabs(t) - (ra + rb);
      NEAR_MISS(sep, margin);
      if (sep > margin) return;
      if (sep > ebest) { ebest = sep; ei = i; ej = j; double sg = t < 0 ? -1 : 1; for (int k = 0; k < 3; k++) en[k] = sg * ax[k]; }
    }
  /* faces are preferred: an edge axis must beat the best face axis by a clear margin */
  {
    if (ei >= 0 && ebest > best + 1e-7 + 0.02 * fabs(best)) {
      /* closest points between the two supporting edges */
      double pa[3], pb[3];
      for (int k = 0; k < 3; k++) { pa[k] = p1[k]; pb[k] = p2[k]; }
      for (int k = 0; k < 3; k++) {
        if (k == ei) continue;
        double sg = dot3(en, A[k]) > 0 ? 1 : -1;
        for (int c = 0; c < 3; c++) pa[c] += sg * a[k] * A[k][c];
      }
      for (int k = 0; k < 3; k++) {
        if (k == ej) continue;
        double sg = dot3(en, Bx[k]) > 0 ? -1 : 1;
        for (int c = 0; c < 3; c++) pb[c] += sg * b[k] * Bx[k][c];
      }
      /* lines pa + s*A[ei], pb + t*B[ej] */
      double w[3] = {pa[0] - pb[0], pa[1] - pb[1], pa[2] - pb[2]};
      double uu = 1, vv = 1, uv = dot3(A[ei], Bx[ej]), uw = dot3(A[ei], w), vw = dot3(Bx[ej], w);
      double den = uu * vv - uv * uv;
      double sc = den > 1e-12 ? (uv * vw - vv * uw) / den : 0, tc = den > 1e-12 ? (uu * vw - uv * uw) / den : 0;
      double ca[3], cb[3], pos[3];
      for (int k = 0; k < 3; k++) { ca[k] = pa[k] + sc * A[ei][k]; cb[k] = pb[k] + tc * Bx[ej][k]; pos[k] = 0.5 * (ca[k] + cb[k]); }
      add_contact(s, pair, pos, en, ebest);
      return;
    }
  }
  /* face contact: reference box = owner of the best face axis */
  int ref1 = code < 3;
  const double* pr = ref1 ? p1 : p2; const double* pi = ref1 ? p2 : p1;
  double (*Ar)[3] = ref1 ? A : Bx; double (*Ai)[3] = ref1 ? Bx : A;
  const double* sr = ref1 ? a : b; const double* si = ref1 ? b : a;
  int ax = ref1 ? code : code - 3;
  /* outward normal of the reference face, pointing to the incident box */
  double nr[3]; for (int k = 0; k < 3; k++) nr[k] = ref1 ? bn[k] : -bn[k];
  /* incident face: the face of the other box most anti-parallel to nr */
  int iax = 0; double mind = 1e30, isg = 1;
  for (int k = 0; k < 3; k++) {
    double dd = dot3(Ai[k], nr);
    if (dd < mind) { mind = dd; iax = k; isg = 1; }
    if (-dd < mind) { mind = -dd; iax = k; isg = -1; }
  }
  int u = (iax + 1) % 3, v = (iax + 2) % 3;
  double fc[3]; for (int k = 0; k < 3; k++) fc[k] = pi[k] + isg * si[iax] * Ai[iax][k];
  double quad[4][3];
  for (int c = 0; c < 4; c++) {
    double su = (c == 0 || c == 3) ? 1 : -1, sv = (c < 2) ? 1 : -1;
    for (int k = 0; k < 3; k++) quad[c][k] = fc[k] + su * si[u] * Ai[u][k] + sv * si[v] * Ai[v][k];
  }
  /* express in the reference face frame (ru, rv in-plane, nr normal through face centre) */
  int ru = (ax + 1) % 3, rv = (ax + 2) % 3;
  double rc[3]; for (int k = 0; k < 3; k++) rc[k] = pr[k] + sr[ax] * nr[k];
  double poly[16][2], hq[4];
  for (int c = 0; c < 4; c++) {
    double w[3] = {quad[c][0] - rc[0], quad[c][1] - rc[1], quad[c][2] - rc[2]};
    poly[c][0] = dot3(w, Ar[ru]); poly[c][1] = dot3(w, Ar[rv]); hq[c] = dot3(w, nr);
  }
  /* plane of the incident face in (u,v,h): h = h0 + gu*x + gv*y ; solve from 3 corners */
  double x0 = poly[0][0], y0 = poly[0][1], x1 = poly[1][0] - x0, y1 = poly[1][1] - y0, x2 = poly[3][0] - x0, y2 = poly[3][1] - y0;
  double h1 = hq[1] - hq[0], h2 = hq[3] - hq[0];
  double det = x1 * y2 - x2 * y1, gu = 0, gv = 0;
  if (fabs(det) > 1e-14) { gu = (h1 * y2 - h2 * y1) / det; gv = (x1 * h2 - x2 * h1) / det; }
  int n = 4;
  n = clip_poly(poly, n, 0, sr[ru], 1); if (n) n = clip_poly(poly, n, 0, sr[ru], -1);
  if (n) n = clip_poly(poly, n, 1, sr[rv], 1); if (n) n = clip_poly(poly, n, 1, sr[rv], -1);
  double nrm[3] = {bn[0], bn[1], bn[2]};
  int cnt = 0;
  for (int c = 0; c < n && cnt < 8; c++) {
    double h = hq[0] + gu * (poly[c][0] - x0) + gv * (poly[c][1] - y0);
    if (fabs(det) <= 1e-14) h = hq[0];
    NEAR_MISS(h, margin);
    if (h > margin) continue;
    /* skip duplicates */
    int dup = 0;
    for (int e = 0; e < c; e++) if (fabs(poly[e][0] - poly[c][0]) + fabs(poly[e][1] - poly[c][1]) < 1e-12) dup = 1;
    if (dup) continue;
    double pos[3];
    for (int k = 0; k < 3; k++) pos[k] = rc[k] + poly[c][0] * Ar[ru][k] + poly[c][1] * Ar[rv][k] + 0.5 * h * nr[k];
    add_contact(s, pair, pos, nrm, h); cnt++;
  }
}


/* ------------------------------------------------------------------ general convex pairs: Minkowski Portal Refinement
 * MuJoCo (up to 3.1; later versions keep it behind the "nativeccd" disable flag) sends every pair that involves an ellipsoid, a
 * cylinder or a mesh -- other than against a plane -- to libccd's ccdMPRPenetration [3P, not in /root/reference], with
 *   centre(geom)      = geom_xpos,
 *   support(geom, d)  = the geom's farthest point along d, plus d * margin/2 (each geom inflated by half the pair margin),
 *   tolerance         = opt.mpr_tolerance (1e-6),  iteration cap = opt.mpr_iterations (50),
 * and turns the result into ONE contact: dist = margin - depth, normal = the returned direction (geom1 -> geom2), pos = the
 * returned position.  This is a restatement of the published algorithm (G. Snethen, "XenoCollide: Complex Collision Made Simple",
 * Game Programming Gems 7; the structure of libccd's mpr.c: discover portal -> refine portal -> find penetration), written from
 * its description; the normal of contacts with smooth geoms is then replaced by the analytic one (smooth_normal below).
 * PARITY UNPINNED like the rest of this file. */
#ifndef MPR_EPS   /* libccd's CCD_EPS in MuJoCo's double-precision build; -DMPR_EPS=1.1920929e-7 reproduces what round 2's device build did (tools/emu_tolerances.py) */
#define MPR_EPS 2.220446049250313e-16
#endif
typedef struct { double v[3], v1[3], v2[3]; } mpr_pt;
static int mpr_zero(double x) { return fabs(x) < MPR_EPS; }
static int mpr_eq(double a, double b) {
  double ab = fabs(a - b);
  if (ab < MPR_EPS) return 1;
  a = fabs(a); b = fabs(b);
  return ab < MPR_EPS * (b > a ? b : a);
}
static double sgn1(double x) { return x > 0 ? 1.0 : (x < 0 ? -1.0 : 0.0); }
static void normalize3(double* v) { double n = norm3(v); if (n > 0) { v[0] /= n; v[1] /= n; v[2] /= n; } }

/* farthest point of geom g along the world direction d (unit), inflated by hm along d */
static void geom_support(const orc_sim* s, int g, const double* d, double hm, double* out) {
  const grx_model_view* m = &s->m;
  const double* R = s->geom_xmat + 9 * g; const double* pos = s->geom_xpos + 3 * g; const double* sz = m->geom_size + 3 * g;
  double dl[3], r[3] = {0, 0, 0};
  mulMatTVec3(dl, R, d);
  switch (m->geom_type[g]) {
    case GRX_GEOM_SPHERE: for (int k = 0; k < 3; k++) r[k] = dl[k] * sz[0]; break;
    case GRX_GEOM_CAPSULE: for (int k = 0; k < 3; k++) r[k] = dl[k] * sz[0]; r[2] += sgn1(dl[2]) * sz[1]; break;
    case GRX_GEOM_ELLIPSOID: {
      double t[3] = {dl[0] * sz[0], dl[1] * sz[1], dl[2] * sz[2]};
      normalize3(t);
      for (int k = 0; k < 3; k++) r[k] = t[k] * sz[k];
      break;
    }
    case GRX_GEOM_CYLINDER: {
      double h = sqrt(dl[0] * dl[0] + dl[1] * dl[1]);
      if (h > MINVAL) { r[0] = dl[0] / h * sz[0]; r[1] = dl[1] / h * sz[0]; }
      r[2] = sgn1(dl[2]) * sz[1];
      break;
    }
    case GRX_GEOM_BOX: for (int k = 0; k < 3; k++) r[k] = sgn1(dl[k]) * sz[k]; break;
    case GRX_GEOM_MESH: {   /* convex hull of the mesh: the hull vertex farthest along d (exhaustive; first maximum wins ties) */
      const int adr = m->geom_hulladr[g], num = m->geom_hullnum[g];
      double best = -1e300; int bi = 0;
      for (int v = 0; v < num; v++) { double t = dot3(m->mesh_vert + 3 * (adr + v), dl); if (t > best) { best = t; bi = v; } }
      if (num > 0) for (int k = 0; k < 3; k++) r[k] = m->mesh_vert[3 * (adr + bi) + k];
      break;
    }
    default: break;
  }
  mulMatVec3(out, R, r);
  for (int k = 0; k < 3; k++) out[k] += pos[k] + d[k] * hm;
}
static long g_mpr_support_calls = 0;
long orc_mpr_support_calls(void) { return g_mpr_support_calls; }
static void mpr_support(const orc_sim* s, int g1, int g2, const double* d, double hm, mpr_pt* o) {
  g_mpr_support_calls++;
  double nd[3] = {-d[0], -d[1], -d[2]};
  geom_support(s, g1, d, hm, o->v1); geom_support(s, g2, nd, hm, o->v2);
  for (int k = 0; k < 3; k++) o->v[k] = o->v1[k] - o->v2[k];
}
static void mpr_portal_dir(const mpr_pt* P, double* dir) {
  double a[3], b[3];
  for (int k = 0; k < 3; k++) { a[k] = P[2].v[k] - P[1].v[k]; b[k] = P[3].v[k] - P[1].v[k]; }
  cross3(dir, a, b); normalize3(dir);
}
static int mpr_reach_tolerance(const mpr_pt* P, const mpr_pt* v4, const double* dir, double tol) {
  double d4 = dot3(v4->v, dir), m1 = d4 - dot3(P[1].v, dir), m2 = d4 - dot3(P[2].v, dir), m3 = d4 - dot3(P[3].v, dir);
  double mn = fmin(m1, fmin(m2, m3));
  return mpr_eq(mn, tol) || mn < tol;
}
static void mpr_expand(mpr_pt* P, const mpr_pt* v4) {
  double c[3];
  cross3(c, v4->v, P[0].v);
  if (dot3(P[1].v, c) > 0) { if (dot3(P[2].v, c) > 0) P[1] = *v4; else P[3] = *v4; }
  else { if (dot3(P[3].v, c) > 0) P[2] = *v4; else P[1] = *v4; }
}
/* squared distance origin -> segment [a,b], witness w */
static double mpr_seg_dist2(const double* a, const double* b, double* w) {
  double d[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, t = -dot3(a, d), dd = dot3(d, d);
  t = dd > 0 ? fmin(1.0, fmax(0.0, t / dd)) : 0.0;
  for (int k = 0; k < 3; k++) w[k] = a[k] + t * d[k];
  return dot3(w, w);
}
/* squared distance origin -> triangle (x0,b,c), witness w (closest point) */
static double mpr_tri_dist2(const double* x0, const double* b, const double* c, double* w) {
  double d1[3], d2[3];
  for (int k = 0; k < 3; k++) { d1[k] = b[k] - x0[k]; d2[k] = c[k] - x0[k]; }
  double u = dot3(x0, x0), v = dot3(d1, d1), ww = dot3(d2, d2), p = dot3(x0, d1), q = dot3(x0, d2), r = dot3(d1, d2);
  double den = ww * v - r * r, best;
  if (!mpr_zero(den)) {
    double sp = (q * r - ww * p) / den, tp = (-sp * r - q) / ww;
    if ((mpr_zero(sp) || sp > 0) && (mpr_eq(sp, 1) || sp < 1) && (mpr_zero(tp) || tp > 0) && (mpr_eq(tp, 1) || tp < 1) && (mpr_eq(tp + sp, 1) || tp + sp < 1)) {
      for (int k = 0; k < 3; k++) w[k] = x0[k] + sp * d1[k] + tp * d2[k];
      best = sp * sp * v + tp * tp * ww + 2 * sp * tp * r + 2 * sp * p + 2 * tp * q + u;
      return best > 0 ? best : 0;
    }
  }
  double w2[3], dist;
  best = mpr_seg_dist2(x0, b, w);
  dist = mpr_seg_dist2(x0, c, w2); if (dist < best) { best = dist; memcpy(w, w2, sizeof(w2)); }
  dist = mpr_seg_dist2(b, c, w2); if (dist < best) { best = dist; memcpy(w, w2, sizeof(w2)); }
  return best;
}
/* 0 = penetration found: depth, dir, pos as libccd defines them, plus the points w1 / w2 of the two geoms' surfaces that the final
 * portal triangle is made of, taken at the foot of the origin on the portal plane;  -1 = separated */
static int mpr_penetration(const orc_sim* s, int g1, int g2, double hm, double tol, int maxit, double* depth, double* dir, double* pos, double* w1, double* w2) {
  mpr_pt P[4], v4;
  double d[3], a[3], b[3], dotv;
  /* portal discovery: P0 = interior point (difference of the centres) */
  for (int k = 0; k < 3; k++) { P[0].v1[k] = s->geom_xpos[3 * g1 + k]; P[0].v2[k] = s->geom_xpos[3 * g2 + k]; P[0].v[k] = P[0].v1[k] - P[0].v2[k]; }
  if (mpr_eq(P[0].v[0], 0) && mpr_eq(P[0].v[1], 0) && mpr_eq(P[0].v[2], 0)) P[0].v[0] += MPR_EPS * 10;
  for (int k = 0; k < 3; k++) d[k] = -P[0].v[k];
  normalize3(d);
  mpr_support(s, g1, g2, d, hm, &P[1]);
  dotv = dot3(P[1].v, d);
  if (mpr_zero(dotv) || dotv < 0) return -1;
  cross3(d, P[0].v, P[1].v);
  if (mpr_zero(dot3(d, d))) {
    /* the origin lies on the ray P0 -> P1 */
    for (int k = 0; k < 3; k++) { w1[k] = P[1].v1[k]; w2[k] = P[1].v2[k]; pos[k] = 0.5 * (w1[k] + w2[k]); }
    if (mpr_eq(P[1].v[0], 0) && mpr_eq(P[1].v[1], 0) && mpr_eq(P[1].v[2], 0)) { *depth = 0; dir[0] = dir[1] = dir[2] = 0; return 0; }   /* touching */
    memcpy(dir, P[1].v, 3 * sizeof(double)); *depth = norm3(dir); normalize3(dir);
    return 0;
  }
  normalize3(d);
  mpr_support(s, g1, g2, d, hm, &P[2]);
  dotv = dot3(P[2].v, d);
  if (mpr_zero(dotv) || dotv < 0) return -1;
  for (int k = 0; k < 3; k++) { a[k] = P[1].v[k] - P[0].v[k]; b[k] = P[2].v[k] - P[0].v[k]; }
  cross3(d, a, b); normalize3(d);
  if (dot3(d, P[0].v) > 0) { mpr_pt t = P[1]; P[1] = P[2]; P[2] = t; d[0] = -d[0]; d[1] = -d[1]; d[2] = -d[2]; }
  for (int guard = 0;; guard++) {
    if (guard > 200) return -1;
    mpr_support(s, g1, g2, d, hm, &P[3]);
    dotv = dot3(P[3].v, d);
    if (mpr_zero(dotv) || dotv < 0) return -1;
    int cont = 0;
    cross3(a, P[1].v, P[3].v); dotv = dot3(a, P[0].v);
    if (dotv < 0 && !mpr_zero(dotv)) { P[2] = P[3]; cont = 1; }
    if (!cont) {
      cross3(a, P[3].v, P[2].v); dotv = dot3(a, P[0].v);
      if (dotv < 0 && !mpr_zero(dotv)) { P[1] = P[3]; cont = 1; }
    }
    if (!cont) break;
    for (int k = 0; k < 3; k++) { a[k] = P[1].v[k] - P[0].v[k]; b[k] = P[2].v[k] - P[0].v[k]; }
    cross3(d, a, b); normalize3(d);
  }
  /* portal refinement: move the portal outwards until the origin is inside the tetrahedron P0..P3 */
  for (int guard = 0;; guard++) {
    if (guard > 200) return -1;
    mpr_portal_dir(P, d);
    dotv = dot3(d, P[1].v);
    if (mpr_zero(dotv) || dotv > 0) break;
    mpr_support(s, g1, g2, d, hm, &v4);
    dotv = dot3(v4.v, d);
    if (!(mpr_zero(dotv) || dotv > 0) || mpr_reach_tolerance(P, &v4, d, tol)) return -1;
    mpr_expand(P, &v4);
  }
  /* penetration: push the portal to the surface of the Minkowski difference */
  for (int it = 0;; it++) {
    mpr_portal_dir(P, d);
    mpr_support(s, g1, g2, d, hm, &v4);
    if (mpr_reach_tolerance(P, &v4, d, tol) || it > maxit) {
      double w[3];
      *depth = sqrt(mpr_tri_dist2(P[1].v, P[2].v, P[3].v, w));
      if (mpr_zero(w[0]) && mpr_zero(w[1]) && mpr_zero(w[2])) memcpy(w, d, sizeof(w));
      normalize3(w); memcpy(dir, w, sizeof(w));
      /* position: barycentric coordinates of the origin in the portal tetrahedron, applied to the two witness sets */
      double bc[4], c[3], sum;
      cross3(c, P[1].v, P[2].v); bc[0] = dot3(c, P[3].v);
      cross3(c, P[3].v, P[2].v); bc[1] = dot3(c, P[0].v);
      cross3(c, P[0].v, P[1].v); bc[2] = dot3(c, P[3].v);
      cross3(c, P[2].v, P[1].v); bc[3] = dot3(c, P[0].v);
      sum = bc[0] + bc[1] + bc[2] + bc[3];
      if (mpr_zero(sum) || sum < 0) {
        bc[0] = 0;
        cross3(c, P[2].v, P[3].v); bc[1] = dot3(c, d);
        cross3(c, P[3].v, P[1].v); bc[2] = dot3(c, d);
        cross3(c, P[1].v, P[2].v); bc[3] = dot3(c, d);
        sum = bc[1] + bc[2] + bc[3];
      }
      for (int k = 0; k < 3; k++) {
        double p1 = 0, p2 = 0;
        for (int i = 0; i < 4; i++) { p1 += bc[i] * P[i].v1[k]; p2 += bc[i] * P[i].v2[k]; }
        pos[k] = 0.5 * (p1 + p2) / sum;
      }
      /* surface witnesses: barycentric weights of the foot of the origin in the portal triangle alone (no interior point) */
      cross3(c, P[2].v, P[3].v); bc[1] = dot3(c, d);
      cross3(c, P[3].v, P[1].v); bc[2] = dot3(c, d);
      cross3(c, P[1].v, P[2].v); bc[3] = dot3(c, d);
      sum = bc[1] + bc[2] + bc[3];
      for (int k = 0; k < 3; k++) {
        w1[k] = (bc[1] * P[1].v1[k] + bc[2] * P[2].v1[k] + bc[3] * P[3].v1[k]) / sum;
        w2[k] = (bc[1] * P[1].v2[k] + bc[2] * P[2].v2[k] + bc[3] * P[3].v2[k]) / sum;
      }
      return 0;
    }
    mpr_expand(P, &v4);
  }
}
/* Analytic outward normal of a smooth geom (sphere, capsule, ellipsoid) at the world point p; 0 if the geom type has none.
 * For shallow penetrations the portal direction is the normal of a triangle whose size (~sqrt(tolerance * curvature radius)) is
 * comparable to the depth, i.e. ill-conditioned; MuJoCo therefore replaces the normal of contacts that involve smooth geoms by
 * the analytic one evaluated at the contact position (mjc_fixNormal [3P]).  Restated from its description: one smooth geom ->
 * its normal; two -> the normalised sum of geom 1's outward normal and geom 2's inward normal; none -> the portal direction. */
static int smooth_normal(const orc_sim* s, int g, const double* p, double* n) {
  const grx_model_view* m = &s->m;
  const double* R = s->geom_xmat + 9 * g; const double* c = s->geom_xpos + 3 * g; const double* sz = m->geom_size + 3 * g;
  double d[3] = {p[0] - c[0], p[1] - c[1], p[2] - c[2]}, loc[3], nl[3];
  mulMatTVec3(loc, R, d);
  switch (m->geom_type[g]) {
    case GRX_GEOM_SPHERE: memcpy(nl, loc, sizeof(nl)); break;
    case GRX_GEOM_CAPSULE: nl[0] = loc[0]; nl[1] = loc[1]; nl[2] = loc[2] > sz[1] ? loc[2] - sz[1] : (loc[2] < -sz[1] ? loc[2] + sz[1] : 0.0); break;
    case GRX_GEOM_ELLIPSOID: for (int k = 0; k < 3; k++) nl[k] = loc[k] / (sz[k] * sz[k]); break;
    default: return 0;
  }
  double len = norm3(nl);
  if (len < MINVAL) return 0;
  for (int k = 0; k < 3; k++) nl[k] /= len;
  mulMatVec3(n, R, nl);
  return 1;
}
static void collide_convex(orc_sim* s, int pair, int g1, int g2, double margin) {
  const grx_model_view* m = &s->m;
  double depth, dir[3], pos[3], w1[3], w2[3], n1[3], n2[3];
  if (mpr_penetration(s, g1, g2, 0.5 * margin, m->opt[GRX_MPR_TOLERANCE], (int)m->opt[GRX_MPR_ITERATIONS], &depth, dir, pos, w1, w2) != 0) return;
  if (dir[0] == 0 && dir[1] == 0 && dir[2] == 0) return;   /* touching, normal undefined: no contact */
  const int h1 = smooth_normal(s, g1, pos, n1), h2 = smooth_normal(s, g2, pos, n2);
  if (h1 || h2) {
    double n[3] = {(h1 ? n1[0] : 0) - (h2 ? n2[0] : 0), (h1 ? n1[1] : 0) - (h2 ? n2[1] : 0), (h1 ? n1[2] : 0) - (h2 ? n2[2] : 0)};
    double len = norm3(n);
    if (len > MINVAL) {
      for (int k = 0; k < 3; k++) dir[k] = n[k] / len;
      /* Penetration measured along the corrected normal: (point of geom 1 - point of geom 2) . n, where the point of a smooth geom
       * is its extreme point along n and the point of a box / cylinder is the portal witness (it lies on the touching face near the
       * contact; the extreme point of a box would be a far corner, with a lever arm of the box size on any error of n).
       * (The distance to the final portal triangle, which libccd reports, depends on the size of that triangle -- rounding-level
       * decisions of the refinement -- when depth ~ sqrt(tolerance * radius): the regime of every resting hand contact.) */
      double nd[3] = {-dir[0], -dir[1], -dir[2]};
      if (h1) geom_support(s, g1, dir, 0.5 * margin, w1);
      if (h2) geom_support(s, g2, nd, 0.5 * margin, w2);
      depth = (w1[0] - w2[0]) * dir[0] + (w1[1] - w2[1]) * dir[1] + (w1[2] - w2[2]) * dir[2];
    }
  }
  add_contact(s, pair, pos, dir, margin - depth);
}
/* plane vs cylinder (MuJoCo's analytic routine [3P], restated from its description): the rim point of the near cap closest to
 * the plane; the corresponding rim point of the far cap; and, when the near cap is close to parallel, the two other corners of an
 * equilateral triangle inscribed in its rim -- up to four contacts, all with the plane normal. */
static void collide_plane_cylinder(orc_sim* s, int pair, int g1, int g2, double margin) {
  const grx_model_view* m = &s->m;
  const double* pm = s->geom_xmat + 9 * g1; const double* cm = s->geom_xmat + 9 * g2; const double* cp = s->geom_xpos + 3 * g2;
  const double r = m->geom_size[3 * g2], hl = m->geom_size[3 * g2 + 1];
  double n[3] = {pm[2], pm[5], pm[8]}, ax[3] = {cm[2], cm[5], cm[8]};
  double prjaxis = dot3(n, ax);
  if (prjaxis > 0) { ax[0] = -ax[0]; ax[1] = -ax[1]; ax[2] = -ax[2]; prjaxis = -prjaxis; }
  double dd[3] = {cp[0] - s->geom_xpos[3 * g1], cp[1] - s->geom_xpos[3 * g1 + 1], cp[2] - s->geom_xpos[3 * g1 + 2]};
  const double dist0 = dot3(dd, n);
  double vec[3] = {ax[0] * prjaxis - n[0], ax[1] * prjaxis - n[1], ax[2] * prjaxis - n[2]};
  double len2 = dot3(vec, vec);
  if (len2 >= MINVAL * MINVAL) { double sc = r / sqrt(len2); vec[0] *= sc; vec[1] *= sc; vec[2] *= sc; }
  else { vec[0] = cm[0] * r; vec[1] = cm[3] * r; vec[2] = cm[6] * r; }          /* cap parallel to the plane: the cylinder's x axis */
  const double prjvec = dot3(vec, n);
  ax[0] *= hl; ax[1] *= hl; ax[2] *= hl; prjaxis *= hl;
  double dist = dist0 + prjaxis + prjvec, pos[3];
  NEAR_MISS(dist, margin);
  if (dist > margin) return;
  for (int k = 0; k < 3; k++) pos[k] = cp[k] + vec[k] + ax[k] - n[k] * dist * 0.5;
  add_contact(s, pair, pos, n, dist);
  dist = dist0 - prjaxis + prjvec;
  if (dist <= margin) {
    for (int k = 0; k < 3; k++) pos[k] = cp[k] + vec[k] - ax[k] - n[k] * dist * 0.5;
    add_contact(s, pair, pos, n, dist);
  }
  const double prjvec1 = -0.5 * prjvec;
  dist = dist0 + prjaxis + prjvec1;
  if (dist <= margin) {
    double v1[3];
    cross3(v1, vec, ax);
    double l = norm3(v1);
    if (l > 0) { double sc = r * sqrt(3.0) * 0.5 / l; v1[0] *= sc; v1[1] *= sc; v1[2] *= sc; }
    for (int sgn = 1; sgn >= -1; sgn -= 2) {
      for (int k = 0; k < 3; k++) pos[k] = cp[k] + sgn * v1[k] + ax[k] - 0.5 * vec[k] - n[k] * dist * 0.5;
      add_contact(s, pair, pos, n, dist);
    }
  }
}
/* plane vs ellipsoid: the deepest point of the ellipsoid (MuJoCo's analytic plane routine: one contact) */
static void collide_plane_ellipsoid(orc_sim* s, int pair, int g1, int g2, double margin) {
  const grx_model_view* m = &s->m;
  const double* pm = s->geom_xmat + 9 * g1; double n[3] = {pm[2], pm[5], pm[8]}, nd[3] = {-n[0], -n[1], -n[2]}, p[3];
  (void)m;
  geom_support(s, g2, nd, 0.0, p);
  double dd[3] = {p[0] - s->geom_xpos[3 * g1], p[1] - s->geom_xpos[3 * g1 + 1], p[2] - s->geom_xpos[3 * g1 + 2]};
  double dist = dot3(dd, n);
  NEAR_MISS(dist, margin);
  if (dist > margin) return;
  double pos[3] = {p[0] - 0.5 * dist * n[0], p[1] - 0.5 * dist * n[1], p[2] - 0.5 * dist * n[2]};
  add_contact(s, pair, pos, n, dist);
}

/* Separating-axis test of the two geoms' oriented bounding boxes (geom_aabb: centre and half extents in the geom frame), each
 * grown by margin / 2: 1 if they may be closer than `margin` (the 6 face axes and the 9 edge-edge axes; a conservative filter in front of
 * the convex routine, it never changes a result) */
static int obb_overlap(const orc_sim* s, int g1, int g2, double margin) {
  const grx_model_view* m = &s->m;
  const double* R1 = s->geom_xmat + 9 * g1; const double* R2 = s->geom_xmat + 9 * g2;
  double c1[3], c2[3], e1[3], e2[3], t[3], A[3][3], B[3][3];
  mulMatVec3(c1, R1, m->geom_aabb + 6 * g1); mulMatVec3(c2, R2, m->geom_aabb + 6 * g2);
  for (int k = 0; k < 3; k++) {
    c1[k] += s->geom_xpos[3 * g1 + k]; c2[k] += s->geom_xpos[3 * g2 + k]; t[k] = c2[k] - c1[k];
    e1[k] = m->geom_aabb[6 * g1 + 3 + k] + 0.5 * margin; e2[k] = m->geom_aabb[6 * g2 + 3 + k] + 0.5 * margin;
    for (int j = 0; j < 3; j++) { A[k][j] = R1[3 * j + k]; B[k][j] = R2[3 * j + k]; }   /* A[k] = k-th axis of box 1 (column k of R1) */
  }
  for (int i = 0; i < 3; i++) {   /* face axes of box 1 and of box 2 */
    double ra = e1[i], rb = 0, tp = fabs(dot3(t, A[i]));
    for (int j = 0; j < 3; j++) rb += e2[j] * fabs(dot3(A[i], B[j]));
    if (tp > ra + rb) return 0;
    ra = 0; rb = e2[i]; tp = fabs(dot3(t, B[i]));
    for (int j = 0; j < 3; j++) ra += e1[j] * fabs(dot3(B[i], A[j]));
    if (tp > ra + rb) return 0;
  }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double ax[3]; cross3(ax, A[i], B[j]);
      double l = norm3(ax);
      if (l < 1e-9) continue;
      double ra = 0, rb = 0;
      for (int k = 0; k < 3; k++) { ra += e1[k] * fabs(dot3(ax, A[k])); rb += e2[k] * fabs(dot3(ax, B[k])); }
      if (fabs(dot3(t, ax)) > ra + rb + 1e-12 * l) return 0;
    }
  return 1;
}

static void collision(orc_sim* s) {
  const grx_model_view* m = &s->m;
  s->ncon = 0;
  for (int p = 0; p < s->npair; p++) {
    int g1 = m->pair_geom1[p], g2 = m->pair_geom2[p];
    double margin = m->pair_margin[p];
    int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
    /* bounding-sphere (or plane) broad phase */
    if (t1 == GRX_GEOM_PLANE) {
      const double* pm = s->geom_xmat + 9 * g1; double n[3] = {pm[2], pm[5], pm[8]};
      double d[3] = {s->geom_xpos[3 * g2] - s->geom_xpos[3 * g1], s->geom_xpos[3 * g2 + 1] - s->geom_xpos[3 * g1 + 1], s->geom_xpos[3 * g2 + 2] - s->geom_xpos[3 * g1 + 2]};
      if (dot3(d, n) > m->geom_rbound[g2] + margin) continue;
    } else {
      double d[3] = {s->geom_xpos[3 * g2] - s->geom_xpos[3 * g1], s->geom_xpos[3 * g2 + 1] - s->geom_xpos[3 * g1 + 1], s->geom_xpos[3 * g2 + 2] - s->geom_xpos[3 * g1 + 2]};
      if (norm3(d) > m->geom_rbound[g1] + m->geom_rbound[g2] + margin) continue;
    }
    if (!m->pair_supported[p]) { s->unsupported_hits++; continue; }
    if (t2 == GRX_GEOM_MESH && t1 != GRX_GEOM_PLANE) {   /* hull against a primitive or another hull: the general convex routine (libccd in MuJoCo [3P]) */
      if (!obb_overlap(s, g1, g2, margin)) continue;       /* conservative filter on the geoms' oriented bounding boxes */
      s->mesh_candidates++;
      int before = s->ncon;
      collide_convex(s, p, g1, g2, margin);
      s->mesh_contacts += s->ncon - before;
      continue;
    }
    if (t1 == GRX_GEOM_SPHERE && t2 == GRX_GEOM_SPHERE) { collide_sphere_sphere(s, p, g1, g2, margin); continue; }
    if (t1 == GRX_GEOM_SPHERE && t2 == GRX_GEOM_CAPSULE) { collide_sphere_capsule(s, p, g1, g2, margin); continue; }
    if (t1 == GRX_GEOM_PLANE && t2 == GRX_GEOM_SPHERE) collide_plane_sphere(s, p, g1, g2, margin);
    else if (t1 == GRX_GEOM_PLANE && t2 == GRX_GEOM_CAPSULE) collide_plane_capsule(s, p, g1, g2, margin);
    else if (t1 == GRX_GEOM_CAPSULE && t2 == GRX_GEOM_BOX) collide_capsule_box(s, p, g1, g2, margin);
    else if (t1 == GRX_GEOM_CAPSULE && t2 == GRX_GEOM_CAPSULE) collide_capsule_capsule(s, p, g1, g2, margin);
    else if (t1 == GRX_GEOM_SPHERE && t2 == GRX_GEOM_BOX) collide_sphere_box(s, p, g1, g2, margin);
    else if (t1 == GRX_GEOM_PLANE && t2 == GRX_GEOM_BOX) collide_plane_box(s, p, g1, g2, margin);
    else if (t1 == GRX_GEOM_PLANE && t2 == GRX_GEOM_MESH) collide_plane_mesh(s, p, g1, g2, margin);
    else if (t1 == GRX_GEOM_BOX && t2 == GRX_GEOM_BOX) collide_box_box(s, p, g1, g2, margin);
    else if (t1 == GRX_GEOM_PLANE && t2 == GRX_GEOM_ELLIPSOID) collide_plane_ellipsoid(s, p, g1, g2, margin);
    else if (t1 == GRX_GEOM_PLANE && t2 == GRX_GEOM_CYLINDER) collide_plane_cylinder(s, p, g1, g2, margin);
    else if (t1 >= GRX_GEOM_SPHERE && t2 <= GRX_GEOM_BOX && (t1 == GRX_GEOM_ELLIPSOID || t1 == GRX_GEOM_CYLINDER || t2 == GRX_GEOM_ELLIPSOID || t2 == GRX_GEOM_CYLINDER))
      collide_convex(s, p, g1, g2, margin);
    else s->unsupported_hits++;
  }
}

/* ------------------------------------------------------------------ K9 constraint rows */
static double* add_row(orc_sim* s, int type, int id, double pos, double margin, double floss, double diagApprox) {
  if (s->nefc >= MAXEFC) return NULL;
  int i = s->nefc++;
  s->efc_type[i] = type; s->efc_id[i] = id; s->efc_pos[i] = pos; s->efc_margin[i] = margin;
  s->efc_frictionloss[i] = floss; s->efc_diagApprox[i] = diagApprox;
  double* J = s->efc_J + (size_t)i * s->nv; memset(J, 0, sizeof(double) * (size_t)s->nv);
  return J;
}

static void make_constraint(orc_sim* s) {
  const grx_model_view* m = &s->m; int nv = s->nv;
  s->nefc = 0;
  double* jp1 = ALLOC(3 * nv); double* jr1 = ALLOC(3 * nv); double* jp2 = ALLOC(3 * nv); double* jr2 = ALLOC(3 * nv);
  /* equality: weld (data layout anchor(3) relpos(3) relquat(4) torquescale) */
  for (int e = 0; e < s->neq; e++) {
    if (!m->eq_active[e] || m->eq_type[e] != GRX_EQ_WELD) continue;
    int b[2] = {m->eq_obj1[e], m->eq_obj2[e]};
    const double* data = m->eq_data + 11 * e; const double* rel = m->eq_relpose + 14 * e;
    double bx[2][3], bq[2][4], bR[2][9], pos[2][3];
    for (int j = 0; j < 2; j++) { /* pose of the ORIGINAL body inside the fused body */
      double v[3]; mulMatVec3(v, s->xmat + 9 * b[j], rel + 7 * j);
      for (int k = 0; k < 3; k++) bx[j][k] = s->xpos[3 * b[j] + k] + v[k];
      mulQuat(bq[j], s->xquat + 4 * b[j], rel + 7 * j + 3); normalize4(bq[j]); quat2mat(bR[j], bq[j]);
      const double* anchor = data + 3 * (1 - j);
      mulMatVec3(v, bR[j], anchor);
      for (int k = 0; k < 3; k++) pos[j][k] = bx[j][k] + v[k];
    }
    double cpos[6];
    for (int k = 0; k < 3; k++) cpos[k] = pos[0][k] - pos[1][k];
    jac_point(s, b[0], pos[0], jp1, jr1); jac_point(s, b[1], pos[1], jp2, jr2);
    double torquescale = data[10];
    double quat[4], quat1[4] = {bq[1][0], -bq[1][1], -bq[1][2], -bq[1][3]}, quat2[4];
    mulQuat(quat, bq[0], data + 6); mulQuat(quat2, quat1, quat);
    for (int k = 0; k < 3; k++) cpos[3 + k] = torquescale * quat2[1 + k];
    double* rows[6];
    for (int r = 0; r < 6; r++)
      rows[r] = add_row(s, EFC_EQUALITY, e, cpos[r], 0, 0, m->eq_invweight[2 * e + (r >= 3)]);
    for (int d = 0; d < nv; d++) {
      for (int r = 0; r < 3; r++) rows[r][d] = jp1[r * nv + d] - jp2[r * nv + d];
      double axis[4] = {0, jr1[d] - jr2[d], jr1[nv + d] - jr2[nv + d], jr1[2 * nv + d] - jr2[2 * nv + d]}, t1[4], t2[4];
      mulQuat(t1, quat1, axis); mulQuat(t2, t1, quat);
      for (int r = 0; r < 3; r++) rows[3 + r][d] = 0.5 * torquescale * t2[1 + r];
    }
  }
  /* equality: joint (MuJoCo mjEQ_JOINT [3P]: q1 - q1_0 = poly(q2 - q2_0), eq_data = polycoef[5], q1_0, q2_0; obj1 / obj2 = the joints).  One row:
   * residual (q1 - q1_0) - poly(x), J = e_dof1 - poly'(x) e_dof2, diagApprox = the two dofs' invweight0 (eq_invweight).  Rows follow the welds'. */
  for (int e = 0; e < s->neq; e++) {
    if (!m->eq_active[e] || m->eq_type[e] != GRX_EQ_JOINT) continue;
    const double* data = m->eq_data + 11 * e;
    const int j1 = m->eq_obj1[e], j2 = m->eq_obj2[e];
    const double x = s->qpos[m->jnt_qposadr[j2]] - data[6];
    const double poly = data[0] + x * (data[1] + x * (data[2] + x * (data[3] + x * data[4])));
    const double deriv = data[1] + x * (2 * data[2] + x * (3 * data[3] + x * 4 * data[4]));
    double* row = add_row(s, EFC_EQUALITY, e, (s->qpos[m->jnt_qposadr[j1]] - data[5]) - poly, 0, 0, m->eq_invweight[2 * e]);
    if (row) { row[m->jnt_dofadr[j1]] = 1.0; row[m->jnt_dofadr[j2]] += -deriv; }
  }
  s->ne = s->nefc;
  /* dof friction loss */
  for (int d = 0; d < nv; d++)
    if (m->dof_frictionloss[d] > 0) {
      double* J = add_row(s, EFC_FRICTION, d, 0, 0, m->dof_frictionloss[d], m->dof_invweight0[d]);
      J[d] = 1;
    }
  s->nf = s->nefc - s->ne;
  /* joint limits */
  for (int j = 0; j < s->njnt; j++) {
    if (!m->jnt_limited[j] || (m->jnt_type[j] != GRX_JNT_SLIDE && m->jnt_type[j] != GRX_JNT_HINGE)) continue;
    double q = s->qpos[m->jnt_qposadr[j]], margin = m->jnt_margin[j]; int d = m->jnt_dofadr[j];
    for (int side = -1; side <= 1; side += 2) {
      double dist = side * (m->jnt_range[2 * j + (side + 1) / 2] - q);
      /* (de)activation gap of the limit row; an exact 0 (joint reset onto its bound) is the same in any precision */
      /* a limit row that switches while its joint is at rest carries no force either way (aref = -b v - k d r with v = 0 at r = 0: continuous): only a MOVING joint
       * near its bound is an activation boundary -- the kitchen's doors and knobs rest 1e-22 ... 3e-7 off their bounds in every snapshot */
      if (fabs(s->qvel[d]) > 1e-4 && fabs(dist - margin) > 0 && fabs(dist - margin) < s->min_activation_gap) s->min_activation_gap = fabs(dist - margin);
      if (dist < margin) {
        double* J = add_row(s, EFC_LIMIT, j, dist, margin, 0, m->dof_invweight0[d]);
        J[d] = -side;
      }
    }
  }
  /* tendon limits (fixed tendons: length = sum coef * qpos, J = coefficients) */
  s->ntl = 0;
  for (int t = 0; t < m->n_tendon_adr; t++) {
    if (!m->tendon_limited[t]) continue;
    double len = 0, margin = m->tendon_margin[t];
    for (int w = m->tendon_adr[t]; w < m->tendon_adr[t] + m->tendon_num[t]; w++) len += m->wrap_coef[w] * s->qpos[m->wrap_qadr[w]];
    for (int side = -1; side <= 1; side += 2) {
      double dist = side * (m->tendon_range[2 * t + (side + 1) / 2] - len);
      if (fabs(dist - margin) > 0 && fabs(dist - margin) < s->min_activation_gap) s->min_activation_gap = fabs(dist - margin);
      if (dist < margin) {
        double* J = add_row(s, EFC_TLIMIT, t, dist, margin, 0, m->tendon_invweight0[t]);
        if (J) { s->ntl++; for (int w = m->tendon_adr[t]; w < m->tendon_adr[t] + m->tendon_num[t]; w++) J[m->wrap_dof[w]] += -side * m->wrap_coef[w]; }
      }
    }
  }
  s->nl = s->nefc - s->ne - s->nf;
  /* contacts (pyramidal cone) */
  for (int c = 0; c < s->ncon; c++) {
    orc_contact* con = &s->con[c];
    if (fabs(con->dist - con->includemargin) < s->min_activation_gap) s->min_activation_gap = fabs(con->dist - con->includemargin);
    if (con->dist >= con->includemargin) continue;
    int b1 = m->geom_bodyid[con->geom1], b2 = m->geom_bodyid[con->geom2];
    jac_point(s, b1, con->pos, jp1, jr1); jac_point(s, b2, con->pos, jp2, jr2);
    int dim = con->dim;
    /* contact-frame jacobian rows: 0 normal, 1,2 tangents, 3 torsion, 4,5 rolling */
    double* Jc = ALLOC(6 * nv);
    for (int r = 0; r < 3; r++)
      for (int d = 0; d < nv; d++) {
        double vp = 0, vr = 0;
        for (int k = 0; k < 3; k++) {
          vp += con->frame[3 * r + k] * (jp2[k * nv + d] - jp1[k * nv + d]);
          vr += con->frame[3 * r + k] * (jr2[k * nv + d] - jr1[k * nv + d]);
        }
        Jc[r * nv + d] = vp; Jc[(3 + r) * nv + d] = vr;
      }
    double tran = m->geom_invweight0[2 * con->geom1] + m->geom_invweight0[2 * con->geom2];
    double rot = m->geom_invweight0[2 * con->geom1 + 1] + m->geom_invweight0[2 * con->geom2 + 1];
    con->efc_address = s->nefc;
    if (dim == 1) {
      double* J = add_row(s, EFC_CONTACT, c, con->dist, con->includemargin, 0, tran);
      if (J) memcpy(J, Jc, sizeof(double) * (size_t)nv);
    } else {
      for (int k = 1; k < dim; k++) {
        double fri = con->friction[k - 1];
        double dA = tran + fri * fri * (k < 3 ? tran : rot);
        for (int sg = 1; sg >= -1; sg -= 2) {
          double* J = add_row(s, EFC_CONTACT, c, con->dist, con->includemargin, 0, dA);
          if (!J) break;
          for (int d = 0; d < nv; d++) J[d] = Jc[d] + sg * fri * Jc[k * nv + d];
        }
      }
    }
    free(Jc);
  }
  free(jp1); free(jr1); free(jp2); free(jr2);
}

/* impedance d(r) */
static double impedance(const double* solimp_in, double pos) {
  double dmin = fmin(MAXIMP, fmax(MINIMP, solimp_in[0])), dmax = fmin(MAXIMP, fmax(MINIMP, solimp_in[1]));
  double width = fmax(0, solimp_in[2]), mid = fmin(MAXIMP, fmax(MINIMP, solimp_in[3])), power = fmax(1, solimp_in[4]);
  if (dmin == dmax || width <= MINVAL) return 0.5 * (dmin + dmax);
  double x = fabs(pos) / width;
  if (x >= 1) return dmax;
  if (x <= 0) return dmin;
  double y;
  if (power == 1) y = x;
  else if (x <= mid) y = pow(x, power) / pow(mid, power - 1);
  else y = 1 - pow(1 - x, power) / pow(1 - mid, power - 1);
  return dmin + y * (dmax - dmin);
}

static void make_impedance(orc_sim* s) {
  const grx_model_view* m = &s->m;
  double h = m->opt[GRX_TIMESTEP];
  for (int i = 0; i < s->nefc; i++) {
    const double *solref, *solimp; int id = s->efc_id[i];
    switch (s->efc_type[i]) {
      case EFC_EQUALITY: solref = m->eq_solref + 2 * id; solimp = m->eq_solimp + 5 * id; break;
      case EFC_FRICTION: solref = m->dof_solref + 2 * id; solimp = m->dof_solimp + 5 * id; break;
      case EFC_LIMIT: solref = m->jnt_solref + 2 * id; solimp = m->jnt_solimp + 5 * id; break;
      case EFC_TLIMIT: solref = m->tendon_solref + 2 * id; solimp = m->tendon_solimp + 5 * id; break;
      default: solref = s->con[id].solref; solimp = s->con[id].solimp; break;
    }
    double pos = s->efc_pos[i] - s->efc_margin[i];
    double imp = impedance(solimp, pos);
    double dmax = fmin(MAXIMP, fmax(MINIMP, solimp[1]));
    double k, b;
    if (solref[0] > 0) {
      double tc = fmax(solref[0], 2 * h), dr = solref[1]; /* refsafe */
      k = 1 / (dmax * dmax * tc * tc * dr * dr); b = 2 / (dmax * tc);
    } else { k = -solref[0] / (dmax * dmax); b = -solref[1] / dmax; }
    if (s->efc_type[i] == EFC_FRICTION) k = 0;
    s->efc_KBIP[4 * i] = k; s->efc_KBIP[4 * i + 1] = b; s->efc_KBIP[4 * i + 2] = imp; s->efc_KBIP[4 * i + 3] = 0;
    s->efc_R[i] = fmax(MINVAL, (1 - imp) * s->efc_diagApprox[i] / imp);
  }
  /* pyramidal contacts: all rows of a contact share R = 2 mu^2 R[first] */
  double impratio = m->opt[GRX_IMPRATIO];
  for (int c = 0; c < s->ncon; c++) {
    orc_contact* con = &s->con[c];
    if (con->efc_address < 0 || con->dim == 1) continue;
    int a = con->efc_address, nr = 2 * (con->dim - 1);
    double mu = con->friction[0] / sqrt(impratio);
    double Rpy = 2 * mu * mu * s->efc_R[a];
    for (int k = 0; k < nr && a + k < s->nefc; k++) s->efc_R[a + k] = Rpy;
  }
  for (int i = 0; i < s->nefc; i++) s->efc_D[i] = 1 / s->efc_R[i];
}

/* ------------------------------------------------------------------ velocity stage */
static void cross_motion(double* r, const double* v, const double* mvec) {
  double t1[3], t2[3];
  cross3(r, v, mvec);
  cross3(t1, v, mvec + 3); cross3(t2, v + 3, mvec);
  r[3] = t1[0] + t2[0]; r[4] = t1[1] + t2[1]; r[5] = t1[2] + t2[2];
}
static void cross_force(double* r, const double* v, const double* f) {
  double t1[3], t2[3];
  cross3(t1, v, f); cross3(t2, v + 3, f + 3);
  r[0] = t1[0] + t2[0]; r[1] = t1[1] + t2[1]; r[2] = t1[2] + t2[2];
  cross3(r + 3, v, f + 3);
}

static void com_vel(orc_sim* s) {
  const grx_model_view* m = &s->m;
  memset(s->cvel, 0, 6 * sizeof(double));
  for (int i = 1; i < s->nbody; i++) {
    double* cv = s->cvel + 6 * i;
    memcpy(cv, s->cvel + 6 * m->body_parent[i], 6 * sizeof(double));
    int da = m->body_dofadr[i], dn = m->body_dofnum[i];
    if (dn == 0) continue;
    int j = 0;
    while (j < dn) {
      int jt = m->jnt_type[m->dof_jntid[da + j]];
      if (jt == GRX_JNT_FREE) {
        for (int k = 0; k < 3; k++) {
          memset(s->cdof_dot + 6 * (da + j + k), 0, 6 * sizeof(double));
          for (int a = 0; a < 6; a++) cv[a] += s->cdof[6 * (da + j + k) + a] * s->qvel[da + j + k];
        }
        for (int k = 3; k < 6; k++) cross_motion(s->cdof_dot + 6 * (da + j + k), cv, s->cdof + 6 * (da + j + k));
        for (int k = 3; k < 6; k++)
          for (int a = 0; a < 6; a++) cv[a] += s->cdof[6 * (da + j + k) + a] * s->qvel[da + j + k];
        j += 6;
      } else {
        cross_motion(s->cdof_dot + 6 * (da + j), cv, s->cdof + 6 * (da + j));
        for (int a = 0; a < 6; a++) cv[a] += s->cdof[6 * (da + j) + a] * s->qvel[da + j];
        j += 1;
      }
    }
  }
}

/* RNE with qacc = 0: Coriolis, centrifugal and gravity (K5) */
static void rne_bias(orc_sim* s) {
  const grx_model_view* m = &s->m; int nb = s->nbody;
  double* cacc = s->cacc; double* cfrc = s->cfrc;
  memset(cacc, 0, 6 * sizeof(double));
  cacc[3] = -m->opt[GRX_GRAVITY_X]; cacc[4] = -m->opt[GRX_GRAVITY_Y]; cacc[5] = -m->opt[GRX_GRAVITY_Z];
  memset(cfrc, 0, 6 * sizeof(double));
  for (int i = 1; i < nb; i++) {
    double* a = cacc + 6 * i; memcpy(a, cacc + 6 * m->body_parent[i], 6 * sizeof(double));
    int da = m->body_dofadr[i];
    for (int j = 0; j < m->body_dofnum[i]; j++)
      for (int k = 0; k < 6; k++) a[k] += s->cdof_dot[6 * (da + j) + k] * s->qvel[da + j];
    const double* I6 = s->cinert + 36 * i; const double* v = s->cvel + 6 * i;
    double Ia[6], Iv[6], t[6];
    for (int r = 0; r < 6; r++) { Ia[r] = Iv[r] = 0; for (int c = 0; c < 6; c++) { Ia[r] += I6[6 * r + c] * a[c]; Iv[r] += I6[6 * r + c] * v[c]; } }
    cross_force(t, v, Iv);
    for (int k = 0; k < 6; k++) cfrc[6 * i + k] = Ia[k] + t[k];
  }
  for (int i = nb - 1; i > 0; i--) {
    int p = m->body_parent[i];
    if (p > 0) for (int k = 0; k < 6; k++) cfrc[6 * p + k] += cfrc[6 * i + k];
  }
  for (int d = 0; d < s->nv; d++) {
    double v = 0; for (int k = 0; k < 6; k++) v += s->cdof[6 * d + k] * cfrc[6 * m->dof_bodyid[d] + k];
    s->qfrc_bias[d] = v;
  }
}

static void passive(orc_sim* s) {
  const grx_model_view* m = &s->m;
  for (int d = 0; d < s->nv; d++) s->qfrc_passive[d] = -m->dof_damping[d] * s->qvel[d];
  for (int j = 0; j < s->njnt; j++) {
    if (m->jnt_stiffness[j] == 0) continue;
    if (m->jnt_type[j] == GRX_JNT_SLIDE || m->jnt_type[j] == GRX_JNT_HINGE)
      s->qfrc_passive[m->jnt_dofadr[j]] -= m->jnt_stiffness[j] * (s->qpos[m->jnt_qposadr[j]] - m->jnt_springref[j]);
  }
}

static void actuation(orc_sim* s) {
  const grx_model_view* m = &s->m;
  memset(s->qfrc_actuator, 0, sizeof(double) * (size_t)s->nv);
  for (int i = 0; i < s->nu; i++) {
    int j = m->act_trnid[i]; double gear = m->act_gear[i];
    double len = gear * s->qpos[m->jnt_qposadr[j]], vel = gear * s->qvel[m->jnt_dofadr[j]];
    double c = s->ctrl[i];
    if (m->act_ctrllimited[i]) c = fmin(m->act_ctrlrange[2 * i + 1], fmax(m->act_ctrlrange[2 * i], c));
    double gain = m->act_gainprm[3 * i];
    if (m->act_gaintype[i] == 1) gain += m->act_gainprm[3 * i + 1] * len + m->act_gainprm[3 * i + 2] * vel;
    double bias = 0;
    if (m->act_biastype[i] == 1) bias = m->act_biasprm[3 * i] + m->act_biasprm[3 * i + 1] * len + m->act_biasprm[3 * i + 2] * vel;
    double f = gain * c + bias;
    if (m->act_forcelimited[i]) f = fmin(m->act_forcerange[2 * i + 1], fmax(m->act_forcerange[2 * i], f));
    s->act_force[i] = f;
    s->qfrc_actuator[m->jnt_dofadr[j]] += gear * f;
  }
}

/* ------------------------------------------------------------------ K10 constraint solve */
/* cost of the constraint part at jar = J a - aref; fills force and active flags */
static double constraint_update(const orc_sim* s, const double* jar, double* force, int* quad) {
  double cost = 0;
  for (int i = 0; i < s->nefc; i++) {
    double D = s->efc_D[i], R = s->efc_R[i], x = jar[i];
    if (s->efc_type[i] == EFC_EQUALITY) { force[i] = -D * x; cost += 0.5 * D * x * x; quad[i] = 1; }
    else if (s->efc_type[i] == EFC_FRICTION) {
      double f = s->efc_frictionloss[i];
      if (x <= -R * f) { force[i] = f; cost += -0.5 * R * f * f - f * x; quad[i] = 0; }
      else if (x >= R * f) { force[i] = -f; cost += -0.5 * R * f * f + f * x; quad[i] = 0; }
      else { force[i] = -D * x; cost += 0.5 * D * x * x; quad[i] = 1; }
    } else {
      if (x < 0) { force[i] = -D * x; cost += 0.5 * D * x * x; quad[i] = 1; }
      else { force[i] = 0; quad[i] = 0; }
    }
  }
  return cost;
}

static double total_cost(const orc_sim* s, const double* qacc, double* Ma, double* jar, double* force, int* quad) {
  int nv = s->nv;
  for (int i = 0; i < nv; i++) { Ma[i] = 0; for (int j = 0; j < nv; j++) Ma[i] += s->M[i * nv + j] * qacc[j]; }
  for (int i = 0; i < s->nefc; i++) {
    double v = 0; for (int j = 0; j < nv; j++) v += s->efc_J[(size_t)i * nv + j] * qacc[j];
    jar[i] = v - s->efc_aref[i];
  }
  double cost = constraint_update(s, jar, force, quad);
  for (int i = 0; i < nv; i++) cost += 0.5 * (Ma[i] - s->qfrc_smooth[i]) * (qacc[i] - s->qacc_smooth[i]);
  return cost;
}

/* derivative of the cost along the search direction at step alpha, and its curvature */
static void ls_eval(const orc_sim* s, const double* jar, const double* jv, double alpha, double quadGauss1, double quadGauss2,
                    double* d1, double* d2) {
  double g = quadGauss1 + alpha * quadGauss2, hss = quadGauss2;
  for (int i = 0; i < s->nefc; i++) {
    double D = s->efc_D[i], R = s->efc_R[i], x = jar[i] + alpha * jv[i];
    if (s->efc_type[i] == EFC_EQUALITY) { g += D * x * jv[i]; hss += D * jv[i] * jv[i]; }
    else if (s->efc_type[i] == EFC_FRICTION) {
      double f = s->efc_frictionloss[i];
      if (x <= -R * f) g += -f * jv[i];
      else if (x >= R * f) g += f * jv[i];
      else { g += D * x * jv[i]; hss += D * jv[i] * jv[i]; }
    } else if (x < 0) { g += D * x * jv[i]; hss += D * jv[i] * jv[i]; }
  }
  *d1 = g; *d2 = hss;
}

static void solve_newton(orc_sim* s) {
  int nv = s->nv, nefc = s->nefc;
  double* qacc = s->qacc;
  double *Ma = ALLOC(nv), *jar = ALLOC(nefc), *force = ALLOC(nefc), *grad = ALLOC(nv), *search = ALLOC(nv), *Mv = ALLOC(nv),
         *jv = ALLOC(nefc), *Hm = ALLOC(nv * nv), *Lh = ALLOC(nv * nv), *tmp = ALLOC(nv);
  int* quad = (int*)calloc((size_t)(nefc > 0 ? nefc : 1), sizeof(int));
  /* warmstart selection (mj_warmstart [3P]) */
  double cw = total_cost(s, s->qacc_warmstart, Ma, jar, force, quad);
  double cs = total_cost(s, s->qacc_smooth, Ma, jar, force, quad);
  memcpy(qacc, cw < cs ? s->qacc_warmstart : s->qacc_smooth, sizeof(double) * (size_t)nv);
  double scale = 1.0 / (s->m.opt[GRX_MEANINERTIA] * (nv > 1 ? nv : 1));
  int it, stall = 0;
  double gnorm = 0, prev_cost = 1e300;
  for (it = 0; it < 100; it++) {
    double cost = total_cost(s, qacc, Ma, jar, force, quad);
    if (prev_cost - cost <= 1e-15 * fabs(cost)) { if (++stall >= 2) break; } else stall = 0;
    prev_cost = cost;
    for (int i = 0; i < nv; i++) {
      double v = Ma[i] - s->qfrc_smooth[i];
      for (int r = 0; r < nefc; r++) v -= s->efc_J[(size_t)r * nv + i] * force[r];
      grad[i] = v;
    }
    gnorm = 0; for (int i = 0; i < nv; i++) gnorm += grad[i] * grad[i];
    gnorm = sqrt(gnorm);
    if (scale * gnorm < 1e-14) break;
    memcpy(Hm, s->M, sizeof(double) * (size_t)(nv * nv));
    for (int r = 0; r < nefc; r++) {
      if (!quad[r]) continue;
      const double* J = s->efc_J + (size_t)r * nv; double D = s->efc_D[r];
      for (int i = 0; i < nv; i++) { if (J[i] == 0) continue; for (int j = 0; j < nv; j++) Hm[i * nv + j] += D * J[i] * J[j]; }
    }
    if (chol_reverse(Lh, Hm, nv)) { s->bad_state |= 2; break; }
    for (int i = 0; i < nv; i++) search[i] = -grad[i];
    chol_reverse_solve(Lh, nv, search);
    for (int i = 0; i < nv; i++) { Mv[i] = 0; for (int j = 0; j < nv; j++) Mv[i] += s->M[i * nv + j] * search[j]; }
    for (int r = 0; r < nefc; r++) { double v = 0; for (int j = 0; j < nv; j++) v += s->efc_J[(size_t)r * nv + j] * search[j]; jv[r] = v; }
    double q1 = 0, q2 = 0;
    for (int i = 0; i < nv; i++) { q1 += search[i] * (Ma[i] - s->qfrc_smooth[i]); q2 += search[i] * Mv[i]; }
    /* exact line search: root of the monotone piecewise-linear derivative */
    double lo = 0, hi = 1, d1, d2, dlo, dhi;
    ls_eval(s, jar, jv, 0, q1, q2, &dlo, &d2);
    if (dlo >= 0) break; /* not a descent direction: converged to rounding */
    ls_eval(s, jar, jv, hi, q1, q2, &dhi, &d2);
    int guard = 0;
    while (dhi < 0 && guard++ < 60) { lo = hi; dlo = dhi; hi *= 2; ls_eval(s, jar, jv, hi, q1, q2, &dhi, &d2); }
    double alpha = hi;
    if (dhi >= 0) {
      alpha = lo;
      ls_eval(s, jar, jv, alpha, q1, q2, &d1, &d2);
      for (int k = 0; k < 200; k++) {
        if (fabs(d1) < 1e-15 * (fabs(q1) + 1e-300)) break;
        double na = alpha - d1 / d2;
        if (!(na > lo && na < hi)) na = 0.5 * (lo + hi);
        alpha = na;
        ls_eval(s, jar, jv, alpha, q1, q2, &d1, &d2);
        if (d1 < 0) lo = alpha; else hi = alpha;
        if (hi - lo < 1e-16 * fmax(1.0, hi)) break;
      }
    }
    for (int i = 0; i < nv; i++) qacc[i] += alpha * search[i];
    (void)cost;
  }
  s->solver_iter = it; s->solver_gradnorm = gnorm;
  total_cost(s, qacc, Ma, jar, force, quad);
  memcpy(s->efc_force, force, sizeof(double) * (size_t)nefc);
  for (int i = 0; i < nv; i++) {
    double v = 0; for (int r = 0; r < nefc; r++) v += s->efc_J[(size_t)r * nv + i] * force[r];
    s->qfrc_constraint[i] = v;
  }
  free(Ma); free(jar); free(force); free(grad); free(search); free(Mv); free(jv); free(Hm); free(Lh); free(tmp); free(quad);
}

/* Noslip post-solver (MuJoCo option noslip_iterations; Adroit: adroit_assets.xml:3 -- restated from the structure of mj_solNoSlip [3P]):
 * projected Gauss-Seidel on the DUAL problem with the regulariser R removed, over the friction-loss rows (box [-floss, floss]) and the pairs
 * of opposing pyramid edges of every frictional contact (their sum -- the normal force -- is kept, the difference is re-solved); equality,
 * limit and frictionless rows keep the forces of the main solver.  A = J M^-1 J' is formed explicitly here (fp64, plain); the device uses the
 * same updates matrix-free.  Ends with qfrc_constraint = J'f and qacc = qacc_smooth + M^-1 qfrc_constraint. */
static void solve_noslip(orc_sim* s, int maxiter) {
  const grx_model_view* m = &s->m; const int nv = s->nv, nefc = s->nefc;
  if (nefc == 0 || maxiter <= 0) return;
  double* B = ALLOC(nefc * nv);   /* M^-1 J' (one column per row, stored as rows) */
  double* A = ALLOC(nefc * nefc);
  double* b = ALLOC(nefc);
  double* f = s->efc_force;
  for (int r = 0; r < nefc; r++) {
    memcpy(B + (size_t)r * nv, s->efc_J + (size_t)r * nv, sizeof(double) * (size_t)nv);
    chol_reverse_solve(s->L, nv, B + (size_t)r * nv);
  }
  for (int r = 0; r < nefc; r++) {
    double v = 0; for (int j = 0; j < nv; j++) v += s->efc_J[(size_t)r * nv + j] * s->qacc_smooth[j];
    b[r] = v - s->efc_aref[r];
    for (int c = 0; c < nefc; c++) { double a = 0; for (int j = 0; j < nv; j++) a += s->efc_J[(size_t)r * nv + j] * B[(size_t)c * nv + j]; A[(size_t)r * nefc + c] = a; }
  }
  const double scale = 1.0 / (m->opt[GRX_MEANINERTIA] * (nv > 1 ? nv : 1)), tol = m->opt[GRX_NOSLIP_TOLERANCE];
  int iter = 0;
  while (iter < maxiter) {
    double improvement = 0;
    if (iter == 0) for (int i = 0; i < nefc; i++) improvement += 0.5 * f[i] * f[i] * s->efc_R[i];   /* cost change of dropping the regulariser */
    /* dry friction */
    for (int i = s->ne; i < s->ne + s->nf; i++) {
      double res = b[i]; for (int c = 0; c < nefc; c++) res += A[(size_t)i * nefc + c] * f[c];
      const double Aii = A[(size_t)i * nefc + i], old = f[i], fl = s->efc_frictionloss[i];
      double fn = old - res / fmax(MINVAL, Aii);
      fn = fn < -fl ? -fl : (fn > fl ? fl : fn);
      f[i] = fn;
      const double d = fn - old;
      improvement -= 0.5 * d * d * Aii + d * res;
    }
    /* contact friction: pairs of opposing pyramid edges */
    for (int c = 0; c < s->ncon; c++) {
      const orc_contact* con = &s->con[c];
      if (con->efc_address < 0 || con->dim == 1) continue;
      for (int j = con->efc_address; j < con->efc_address + 2 * (con->dim - 1) && j + 1 < nefc; j += 2) {
        double res[2];
        for (int k = 0; k < 2; k++) { res[k] = b[j + k]; for (int q = 0; q < nefc; q++) res[k] += A[(size_t)(j + k) * nefc + q] * f[q]; }
        const double A00 = A[(size_t)j * nefc + j], A01 = A[(size_t)j * nefc + j + 1], A10 = A[(size_t)(j + 1) * nefc + j], A11 = A[(size_t)(j + 1) * nefc + j + 1];
        const double o0 = f[j], o1 = f[j + 1];
        const double bc0 = res[0] - (A00 * o0 + A01 * o1), bc1 = res[1] - (A10 * o0 + A11 * o1);
        const double mid = 0.5 * (o0 + o1);
        const double K1 = A00 + A11 - A01 - A10, K0 = mid * (A00 - A11) + bc0 - bc1;
        if (K1 < MINVAL) { f[j] = f[j + 1] = mid; }
        else {
          const double y = -K0 / K1;
          if (y < -mid) { f[j] = 0; f[j + 1] = 2 * mid; }
          else if (y > mid) { f[j] = 2 * mid; f[j + 1] = 0; }
          else { f[j] = mid + y; f[j + 1] = mid - y; }
        }
        const double d0 = f[j] - o0, d1 = f[j + 1] - o1;
        improvement -= 0.5 * (d0 * (A00 * d0 + A01 * d1) + d1 * (A10 * d0 + A11 * d1)) + d0 * res[0] + d1 * res[1];
      }
    }
    iter++;
    if (improvement * scale < tol) break;
  }
  s->noslip_iter_done = iter;
  for (int i = 0; i < nv; i++) {
    double v = 0; for (int r = 0; r < nefc; r++) v += s->efc_J[(size_t)r * nv + i] * f[r];
    s->qfrc_constraint[i] = v;
  }
  memcpy(s->qacc, s->qfrc_constraint, sizeof(double) * (size_t)nv);
  chol_reverse_solve(s->L, nv, s->qacc);
  for (int i = 0; i < nv; i++) s->qacc[i] += s->qacc_smooth[i];
  free(B); free(A); free(b);
}

/* ------------------------------------------------------------------ forward / step */
static int bad_number(const double* x, int n, double maxval) {
  for (int i = 0; i < n; i++) if (!(x[i] == x[i]) || x[i] > maxval || x[i] < -maxval) return 1;
  return 0;
}

/* distance along the ray (origin p, direction d, both in the zone frame) to a sphere of radius r / a box of half sizes sz, or -1
 * (restating MuJoCo's mju_rayGeom for the two zone types in scope: nearest non-negative root; an origin inside the zone hits) */
static double ray_sphere(const double* p, const double* d, double r) {
  double a = dot3(d, d), b = dot3(d, p), c = dot3(p, p) - r * r, det = b * b - a * c;
  if (det < MINVAL || a < MINVAL) return -1;
  double sq = sqrt(det), x0 = (-b - sq) / a, x1 = (-b + sq) / a;
  return x0 >= 0 ? x0 : (x1 >= 0 ? x1 : -1);
}
static double ray_box(const double* p, const double* d, const double* sz) {
  double best = -1;
  for (int i = 0; i < 3; i++) {
    if (fabs(d[i]) < MINVAL) continue;
    for (int side = -1; side <= 1; side += 2) {
      double t = (side * sz[i] - p[i]) / d[i];
      if (t < 0) continue;
      int j = (i + 1) % 3, k = (i + 2) % 3;
      if (fabs(p[j] + t * d[j]) <= sz[j] && fabs(p[k] + t * d[k]) <= sz[k] && (best < 0 || t < best)) best = t;
    }
  }
  return best;
}

/* touch sensors (MuJoCo mjSENS_TOUCH [3P]): sum of the normal forces of the active contacts that involve the zone's body and
 * whose ray (from the contact point along the contact normal, flipped when the zone's body is the second one) meets the zone */
/* ray against a cylinder zone (radius r, half height h along z; mju_rayGeom semantics [3P]): nearest non-negative hit of the side or a cap, -1 if none */
static double ray_cylinder(const double* p, const double* d, double r, double h) {
  double best = -1;
  double a = d[0] * d[0] + d[1] * d[1], b = d[0] * p[0] + d[1] * p[1], c = p[0] * p[0] + p[1] * p[1] - r * r;
  if (a > MINVAL) {
    double det = b * b - a * c;
    if (det >= 0) {
      double sq = sqrt(det);
      for (int k = 0; k < 2; k++) { double t = (-b + (k ? sq : -sq)) / a; if (t >= 0 && fabs(p[2] + t * d[2]) <= h && (best < 0 || t < best)) best = t; }
    }
  }
  if (fabs(d[2]) > MINVAL)
    for (int side = -1; side <= 1; side += 2) {
      double t = (side * h - p[2]) / d[2], x = p[0] + t * d[0], y = p[1] + t * d[1];
      if (t >= 0 && x * x + y * y <= r * r && (best < 0 || t < best)) best = t;
    }
  return best;
}

static void touch_sensors(orc_sim* s) {
  const grx_model_view* m = &s->m;
  for (int t = 0; t < m->n_touch_body; t++) {
    int b = m->touch_body[t];
    double zp[3], zq[4], zR[9], v[3], val = 0;
    mulMatVec3(v, s->xmat + 9 * b, m->touch_pos + 3 * t);
    for (int k = 0; k < 3; k++) zp[k] = s->xpos[3 * b + k] + v[k];
    mulQuat(zq, s->xquat + 4 * b, m->touch_quat + 4 * t); normalize4(zq); quat2mat(zR, zq);
    for (int c = 0; c < s->ncon; c++) {
      const orc_contact* con = &s->con[c];
      int b1 = m->geom_bodyid[con->geom1], b2 = m->geom_bodyid[con->geom2];
      if (con->efc_address < 0 || (b != b1 && b != b2)) continue;
      int nr = con->dim == 1 ? 1 : 2 * (con->dim - 1);
      double fn = 0;
      for (int k = 0; k < nr && con->efc_address + k < s->nefc; k++) fn += s->efc_force[con->efc_address + k];
      if (fn <= 0) continue;
      double sg = (b == b2) ? -1.0 : 1.0, dw[3] = {sg * con->frame[0], sg * con->frame[1], sg * con->frame[2]};
      double pw[3] = {con->pos[0] - zp[0], con->pos[1] - zp[1], con->pos[2] - zp[2]}, pl[3], dl[3];
      mulMatTVec3(pl, zR, pw); mulMatTVec3(dl, zR, dw);
      double hit = m->touch_type[t] == GRX_GEOM_SPHERE ? ray_sphere(pl, dl, m->touch_size[3 * t])
                 : (m->touch_type[t] == GRX_GEOM_CYLINDER ? ray_cylinder(pl, dl, m->touch_size[3 * t], m->touch_size[3 * t + 1]) : ray_box(pl, dl, m->touch_size + 3 * t));
      if (hit >= 0) val += fn;
    }
    s->touch[t] = val;
  }
}

/* restates mj_forward [3P] */
void orc_forward(orc_sim* s) {
  int nv = s->nv;
  kinematics(s);
  com_pos(s);
  crb_and_factor(s);
  if (chol_reverse(s->L, s->M, nv)) s->bad_state |= 4;
  collision(s);
  make_constraint(s);
  make_impedance(s);
  /* velocity stage */
  com_vel(s);
  passive(s);
  for (int i = 0; i < s->nefc; i++) {
    double v = 0; for (int j = 0; j < nv; j++) v += s->efc_J[(size_t)i * nv + j] * s->qvel[j];
    s->efc_vel[i] = v;
    s->efc_aref[i] = -s->efc_KBIP[4 * i + 1] * v - s->efc_KBIP[4 * i] * s->efc_KBIP[4 * i + 2] * (s->efc_pos[i] - s->efc_margin[i]);
  }
  rne_bias(s);
  actuation(s);
  for (int i = 0; i < nv; i++) {
    s->qfrc_smooth[i] = s->qfrc_passive[i] - s->qfrc_bias[i] + s->qfrc_actuator[i];
    s->qacc_smooth[i] = s->qfrc_smooth[i];
  }
  chol_reverse_solve(s->L, nv, s->qacc_smooth);
  if (s->nefc == 0) {
    memcpy(s->qacc, s->qacc_smooth, sizeof(double) * (size_t)nv);
    memset(s->qfrc_constraint, 0, sizeof(double) * (size_t)nv);
    s->solver_iter = 0;
  } else {
    solve_newton(s);
    if (s->m.dims[GRX_NOSLIP_ITERATIONS] > 0) solve_noslip(s, s->m.dims[GRX_NOSLIP_ITERATIONS]);
  }
  memcpy(s->qacc_warmstart, s->qacc, sizeof(double) * (size_t)nv);
  if (s->nefc == 0) memset(s->efc_force, 0, sizeof(s->efc_force));
  touch_sensors(s);
}

/* restates mj_Euler [3P]: implicit joint damping */
static void euler(orc_sim* s) {
  const grx_model_view* m = &s->m; int nv = s->nv; double h = m->opt[GRX_TIMESTEP];
  double* qacc = ALLOC(nv);
  int anydamp = 0;
  for (int d = 0; d < nv; d++) if (m->dof_damping[d] > 0) anydamp = 1;
  if (anydamp && m->dims[GRX_EULERDAMP]) {
    double* A = ALLOC(nv * nv); double* L = ALLOC(nv * nv);
    memcpy(A, s->M, sizeof(double) * (size_t)(nv * nv));
    for (int d = 0; d < nv; d++) A[d * nv + d] += h * m->dof_damping[d];
    for (int d = 0; d < nv; d++) qacc[d] = s->qfrc_smooth[d] + s->qfrc_constraint[d];
    if (chol_reverse(L, A, nv)) s->bad_state |= 8;
    chol_reverse_solve(L, nv, qacc);
    free(A); free(L);
  } else memcpy(qacc, s->qacc, sizeof(double) * (size_t)nv);
  for (int d = 0; d < nv; d++) s->qvel[d] += h * qacc[d];
  for (int j = 0; j < s->njnt; j++) {
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    if (m->jnt_type[j] == GRX_JNT_FREE) {
      for (int k = 0; k < 3; k++) s->qpos[qa + k] += h * s->qvel[da + k];
      double w[3] = {s->qvel[da + 3], s->qvel[da + 4], s->qvel[da + 5]};
      double n = norm3(w);
      if (n > MINVAL) {
        double ax[3] = {w[0] / n, w[1] / n, w[2] / n}, qr[4], qn[4];
        axisAngle2Quat(qr, ax, h * n);
        mulQuat(qn, s->qpos + qa + 3, qr); normalize4(qn);
        memcpy(s->qpos + qa + 3, qn, sizeof(qn));
      }
    } else s->qpos[qa] += h * s->qvel[da];
  }
  s->time += h;
  free(qacc);
}

/* advance (qpos0, qvel0) by velocity v and acceleration a over step hh into the live state (mj_integratePos semantics) */
static void rk_advance(orc_sim* s, const double* qpos0, const double* qvel0, const double* v, const double* a, double hh) {
  const grx_model_view* m = &s->m;
  for (int d = 0; d < s->nv; d++) s->qvel[d] = qvel0[d] + hh * a[d];
  memcpy(s->qpos, qpos0, sizeof(double) * (size_t)s->nq);
  for (int j = 0; j < s->njnt; j++) {
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    if (m->jnt_type[j] == GRX_JNT_FREE) {
      for (int k = 0; k < 3; k++) s->qpos[qa + k] += hh * v[da + k];
      double w[3] = {v[da + 3], v[da + 4], v[da + 5]}, n = norm3(w);
      if (n > MINVAL) {
        double ax[3] = {w[0] / n, w[1] / n, w[2] / n}, qr[4], qn[4];
        axisAngle2Quat(qr, ax, hh * n); mulQuat(qn, s->qpos + qa + 3, qr); normalize4(qn); memcpy(s->qpos + qa + 3, qn, sizeof(qn));
      }
    } else s->qpos[qa] += hh * v[da];
  }
}

/* restates mj_RungeKutta(m, d, 4) [3P]: classical tableau, a full forward pass per stage */
static void rk4(orc_sim* s) {
  int nq = s->nq, nv = s->nv; double h = s->m.opt[GRX_TIMESTEP];
  static const double A[3] = {0.5, 0.5, 1.0}, B[4] = {1.0 / 6, 1.0 / 3, 1.0 / 3, 1.0 / 6};
  double* q0 = ALLOC(nq); double* v0 = ALLOC(nv); double* Fv = ALLOC(4 * nv); double* Fa = ALLOC(4 * nv);
  memcpy(q0, s->qpos, sizeof(double) * (size_t)nq); memcpy(v0, s->qvel, sizeof(double) * (size_t)nv);
  memcpy(Fv, s->qvel, sizeof(double) * (size_t)nv); memcpy(Fa, s->qacc, sizeof(double) * (size_t)nv);  /* stage 0 = the forward pass already done */
  for (int i = 1; i < 4; i++) {
    rk_advance(s, q0, v0, Fv + (i - 1) * nv, Fa + (i - 1) * nv, A[i - 1] * h);
    orc_forward(s);
    memcpy(Fv + i * nv, s->qvel, sizeof(double) * (size_t)nv); memcpy(Fa + i * nv, s->qacc, sizeof(double) * (size_t)nv);
  }
  double* dv = ALLOC(nv); double* da = ALLOC(nv);
  for (int d = 0; d < nv; d++) for (int j = 0; j < 4; j++) { dv[d] += B[j] * Fv[j * nv + d]; da[d] += B[j] * Fa[j * nv + d]; }
  rk_advance(s, q0, v0, dv, da, h);
  s->time += h;
  free(q0); free(v0); free(Fv); free(Fa); free(dv); free(da);
}

/* restates mj_step(model, data, nstep) [3P] */
void orc_step(orc_sim* s, int nstep) {
  for (int k = 0; k < nstep; k++) {
    if (bad_number(s->qpos, s->nq, 1e10) || bad_number(s->qvel, s->nv, 1e10)) { s->bad_state |= 1; orc_reset_data(s); s->bad_state |= 1; }
    orc_forward(s);
    if (bad_number(s->qacc, s->nv, 1e10)) { s->bad_state |= 1; orc_reset_data(s); s->bad_state |= 1; orc_forward(s); }
    if (s->m.dims[GRX_INTEGRATOR] == 1) rk4(s); else euler(s);
  }
}

/* ------------------------------------------------------------------ accessors for the ctypes harness */
double* orc_ptr(orc_sim* s, const char* name) {
#define P(n) if (!strcmp(name, #n)) return s->n;
  P(qpos) P(qvel) P(ctrl) P(mocap_pos) P(mocap_quat) P(qacc_warmstart) P(xpos) P(xquat) P(xmat) P(xipos) P(geom_xpos)
  P(geom_xmat) P(site_xpos) P(site_xmat) P(subtree_com) P(cdof) P(cdof_dot) P(M) P(cvel) P(qfrc_bias) P(qfrc_passive)
  P(qfrc_actuator) P(qfrc_smooth) P(qacc_smooth) P(qfrc_constraint) P(qacc) P(efc_J) P(efc_pos) P(efc_margin) P(efc_D)
  P(efc_R) P(efc_aref) P(efc_force) P(efc_vel) P(efc_diagApprox) P(touch)
#undef P
  if (!strcmp(name, "time")) return &s->time;
  if (!strcmp(name, "shift")) return s->shift;
  if (!strcmp(name, "min_activation_gap")) return &s->min_activation_gap;
  return NULL;
}
/* mutable model table (e.g. eq_data for reset_mocap_welds, mujoco_utils.py:74-80) */
double* orc_model_ptr(orc_sim* s, const char* name) {
#define GRX_FI(n)
#define GRX_FF(n) if (!strcmp(name, #n)) return (double*)s->m.n;
#include "../include/grx_model_fields.def"
#undef GRX_FI
#undef GRX_FF
  return NULL;
}
int orc_int(orc_sim* s, const char* name) {
  if (!strcmp(name, "ncon")) return s->ncon;
  if (!strcmp(name, "nefc")) return s->nefc;
  if (!strcmp(name, "ne")) return s->ne;
  if (!strcmp(name, "nf")) return s->nf;
  if (!strcmp(name, "nl")) return s->nl;
  if (!strcmp(name, "ntl")) return s->ntl;
  if (!strcmp(name, "solver_iter")) return s->solver_iter;
  if (!strcmp(name, "noslip_iter")) return s->noslip_iter_done;
  if (!strcmp(name, "bad_state")) return s->bad_state;
  if (!strcmp(name, "unsupported_hits")) return s->unsupported_hits;
  if (!strcmp(name, "mesh_candidates")) return (int)s->mesh_candidates;
  if (!strcmp(name, "mesh_contacts")) return (int)s->mesh_contacts;
  if (!strcmp(name, "nv")) return s->nv;
  if (!strcmp(name, "nq")) return s->nq;
  return -1;
}
void orc_set_int(orc_sim* s, const char* name, int v) {
  if (!strcmp(name, "opt_disable_mesh_plane")) s->opt_disable_mesh_plane = v;
  if (!strcmp(name, "bad_state")) s->bad_state = v;
}
/* contact dump: 16 doubles per contact: dist, pos3, normal3, geom1, geom2, dim, efc_address, includemargin */
int orc_contacts(orc_sim* s, double* out, int maxn) {
  int n = s->ncon < maxn ? s->ncon : maxn;
  for (int i = 0; i < n; i++) {
    const orc_contact* c = &s->con[i]; double* o = out + 16 * i;
    o[0] = c->dist; memcpy(o + 1, c->pos, 3 * sizeof(double)); memcpy(o + 4, c->frame, 3 * sizeof(double));
    o[7] = c->geom1; o[8] = c->geom2; o[9] = c->dim; o[10] = c->efc_address; o[11] = c->includemargin;
  }
  return s->ncon;
}
