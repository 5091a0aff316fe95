/* rodent_ref.c -- CPU ORACLE (test infrastructure, NOT the product path).
 *
 * A plain-C restatement, one function per stage, of the per-environment physics step the
 * reference executes through `brax.mjx.pipeline.step` -> `mujoco.mjx.step`
 * [REF Rodent_Env_Brax.py:87,101 -> UP mjx.forward/step], and of the reference env's
 * reset/step/obs arithmetic [REF Rodent_Env_Brax.py:71-162].  The MJX/MuJoCo sources are not
 * in /root/reference and the packages (mujoco 3.1.x-3.2.x, mujoco-mjx, brax 0.10-0.11, unpinned)
 * are not installed, so the arithmetic follows the published algorithm (MuJoCo "Computation"
 * chapter).  PINNED to the reference for the forward pass up to the observation:
 * kinematics, subtree COM, cinert, cvel and actuation agree with the one full vector the reference stores
 * (its own mjx.forward output, [NB Env_step.ipynb cell 8]; tests/test_reference_pin.py), next to the structural
 * known-answers of the notebooks (tests/test_known_answers.py).  PARITY UNPINNED for everything past it --
 * mass matrix, bias forces, contacts (incl. the sphere / capsule primitives, restated from memory), the solver,
 * the integrator: the reference holds no step() output; those stages rest on oracle/np_ref.py's independent formulation.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Build: gcc -O2 -shared -fPIC [-DREF_F32] [-fopenmp] rodent_ref.c -lm
 *   REF_F32 selects float arithmetic (calibrates float32 round-off); default is double.
 * All API arrays are double regardless of the internal arithmetic type.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef REF_F32
typedef float real;
#define R(x) x##f
#define SQRT sqrtf
#define FABS fabsf
#define POW powf
#define SIN sinf
#define COS cosf
#define EXP expf
#define FMAX fmaxf
#define FMIN fminf
#else
typedef double real;
#define R(x) x
#define SQRT sqrt
#define FABS fabs
#define POW pow
#define SIN sin
#define COS cos
#define EXP exp
#define FMAX fmax
#define FMIN fmin
#endif

#define MINVAL R(1e-15)
#define MINIMP R(0.0001)
#define MAXIMP R(0.9999)
enum { JNT_FREE = 0, JNT_HINGE = 3 };
/* con_kind (rodent_amd/mjcf.py _collision_tables): plane - sphere / capsule end / ellipsoid, then the pairs of two moving geoms */
enum { CON_SPHERE = 0, CON_CAP_POS = 1, CON_CAP_NEG = 2, CON_ELLIPSOID = 3, CON_SPHERE_SPHERE = 4, CON_SPHERE_CAPSULE = 5, CON_CAPSULE_CAPSULE = 6 };

/* ------------------------------------------------------------------ model blob */
typedef struct {
  char name[32];
  int dtype, ndim, dims[4];
  size_t count;
  void* data; /* float* or int32_t* */
} blob_entry;

typedef struct ref_model {
  int nentries;
  blob_entry* e;
  unsigned char* raw;
  /* dims */
  int nq, nv, nu, na, nbody, njnt, ngeom, nM, ncon, nlimit, nefc, obs_dim;
  int iterations, ls_iterations, solver; /* solver: 1 = CG, 2 = Newton (mjtSolver) */
  real timestep, gravity[3], tolerance, ls_tolerance, impratio, meaninertia;
  /* int tables (borrowed from blob) */
  const int32_t *body_parentid, *body_rootid, *body_jntadr, *body_jntnum, *body_dofadr, *body_dofnum, *body_lastdof;
  const int32_t *jnt_type, *jnt_qposadr, *jnt_dofadr, *jnt_bodyid, *jnt_limited;
  const int32_t *dof_bodyid, *dof_jntid, *dof_parentid, *dof_Madr, *dof_depth;
  const int32_t *geom_bodyid, *con_geom1, *con_geom2, *con_kind, *con_dim, *con_body1, *con_body2, *limit_jnt;
  const int32_t *actuator_qposadr, *actuator_dofadr;
  const int32_t *actuator_momentadr, *actuator_moment_qposadr, *actuator_moment_dofadr; /* sparse transmission (joint: 1 entry, fixed tendon: its joints) */
  /* real tables (converted copies) */
  real *body_pos, *body_quat, *body_ipos, *body_iquat, *body_mass, *body_inertia;
  real *jnt_pos, *jnt_axis, *jnt_stiffness, *jnt_range, *jnt_solref, *jnt_solimp;
  real *dof_armature, *dof_damping, *dof_invweight0, *qpos0, *qpos_spring;
  real *geom_pos, *geom_quat, *geom_size;
  real *con_friction, *con_solref, *con_solimp, *con_invweight;
  real *gain0, *biasprm, *tau, *ctrlrange, *actuator_moment_coef;
} ref_model;

static const blob_entry* find(const ref_model* m, const char* name) {
  for (int i = 0; i < m->nentries; i++)
    if (!strncmp(m->e[i].name, name, 32)) return &m->e[i];
  fprintf(stderr, "rodent_ref: model field '%s' missing\n", name);
  abort();
}
static const int32_t* itab(const ref_model* m, const char* name) { return (const int32_t*)find(m, name)->data; }
static int iscalar(const ref_model* m, const char* name) { return itab(m, name)[0]; }
static real* rtab(const ref_model* m, const char* name) {
  const blob_entry* e = find(m, name);
  real* out = (real*)malloc(sizeof(real) * (e->count ? e->count : 1));
  for (size_t i = 0; i < e->count; i++) out[i] = (real)((const float*)e->data)[i];
  return out;
}

static void unit_n(real* v, int n) {
  real s = 0;
  for (int i = 0; i < n; i++) s += v[i] * v[i];
  s = SQRT(s);
  if (s > MINVAL) for (int i = 0; i < n; i++) v[i] /= s;
}

ref_model* ref_model_load(const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) return NULL;
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  unsigned char* raw = (unsigned char*)malloc(n);
  if (fread(raw, 1, n, f) != (size_t)n) { fclose(f); free(raw); return NULL; }
  fclose(f);
  if (memcmp(raw, "RRM1", 4)) { free(raw); return NULL; }
  ref_model* m = (ref_model*)calloc(1, sizeof(ref_model));
  m->raw = raw;
  m->nentries = *(uint32_t*)(raw + 4);
  m->e = (blob_entry*)calloc(m->nentries, sizeof(blob_entry));
  unsigned char* p = raw + 8;
  for (int i = 0; i < m->nentries; i++, p += 72) {
    blob_entry* e = &m->e[i];
    memcpy(e->name, p, 32);
    e->dtype = *(uint32_t*)(p + 32);
    e->ndim = *(uint32_t*)(p + 36);
    for (int k = 0; k < 4; k++) e->dims[k] = *(uint32_t*)(p + 40 + 4 * k);
    uint64_t off = *(uint64_t*)(p + 56), nb = *(uint64_t*)(p + 64);
    e->count = nb / 4;
    e->data = raw + off;
  }
  m->nq = iscalar(m, "nq"); m->nv = iscalar(m, "nv"); m->nu = iscalar(m, "nu"); m->na = iscalar(m, "na");
  m->nbody = iscalar(m, "nbody"); m->njnt = iscalar(m, "njnt"); m->ngeom = iscalar(m, "ngeom");
  m->nM = iscalar(m, "nM"); m->ncon = iscalar(m, "ncon"); m->nlimit = iscalar(m, "nlimit");
  m->nefc = iscalar(m, "nefc"); m->obs_dim = iscalar(m, "obs_dim");
  m->iterations = iscalar(m, "opt_iterations"); m->ls_iterations = iscalar(m, "opt_ls_iterations");
  m->solver = iscalar(m, "opt_solver");
  real* t;
  t = rtab(m, "opt_timestep"); m->timestep = t[0]; free(t);
  t = rtab(m, "opt_gravity"); memcpy(m->gravity, t, 3 * sizeof(real)); free(t);
  t = rtab(m, "opt_tolerance"); m->tolerance = t[0]; free(t);
  t = rtab(m, "opt_ls_tolerance"); m->ls_tolerance = t[0]; free(t);
  t = rtab(m, "opt_impratio"); m->impratio = t[0]; free(t);
  t = rtab(m, "stat_meaninertia"); m->meaninertia = t[0]; free(t);
#define IT(x) m->x = itab(m, #x)
  IT(body_parentid); IT(body_rootid); IT(body_jntadr); IT(body_jntnum); IT(body_dofadr); IT(body_dofnum); IT(body_lastdof);
  IT(jnt_type); IT(jnt_qposadr); IT(jnt_dofadr); IT(jnt_bodyid); IT(jnt_limited);
  IT(dof_bodyid); IT(dof_jntid); IT(dof_parentid); IT(dof_Madr); IT(dof_depth);
  IT(geom_bodyid); IT(con_geom1); IT(con_geom2); IT(con_kind); IT(con_dim); IT(con_body1); IT(con_body2); IT(limit_jnt);
  IT(actuator_qposadr); IT(actuator_dofadr);
  IT(actuator_momentadr); IT(actuator_moment_qposadr); IT(actuator_moment_dofadr);
#define RT(x) m->x = rtab(m, #x)
  RT(body_pos); RT(body_quat); RT(body_ipos); RT(body_iquat); RT(body_mass); RT(body_inertia);
  RT(jnt_pos); RT(jnt_axis); RT(jnt_stiffness); RT(jnt_range); RT(jnt_solref); RT(jnt_solimp);
  RT(dof_armature); RT(dof_damping); RT(dof_invweight0); RT(qpos0); RT(qpos_spring);
  RT(geom_pos); RT(geom_quat); RT(geom_size);
  RT(con_friction); RT(con_solref); RT(con_solimp); RT(con_invweight); RT(actuator_moment_coef);
  /* unit quaternions / axes are unit only to float32 round-off in the blob: renormalise them in the oracle's arithmetic
   * (MuJoCo normalises them in double at compile time), so that formulations that treat a non-unit vector differently
   * (oracle/np_ref.py) agree to double round-off */
  for (int b = 0; b < m->nbody; b++) { unit_n(m->body_quat + 4 * b, 4); unit_n(m->body_iquat + 4 * b, 4); }
  for (int g = 0; g < m->ngeom; g++) unit_n(m->geom_quat + 4 * g, 4);
  for (int j = 0; j < m->njnt; j++) unit_n(m->jnt_axis + 3 * j, 3);
  m->gain0 = rtab(m, "actuator_gainprm0"); m->biasprm = rtab(m, "actuator_biasprm");
  m->tau = rtab(m, "actuator_dynprm0"); m->ctrlrange = rtab(m, "actuator_ctrlrange");
  return m;
}

void ref_model_set_solver(ref_model* m, int solver) { m->solver = solver; }  /* 1 = CG, 2 = Newton */
void ref_model_set_iterations(ref_model* m, int iterations, int ls_iterations) {
  m->iterations = iterations; m->ls_iterations = ls_iterations;
}

int ref_model_dim(const ref_model* m, const char* name) { return iscalar(m, name); }

/* ------------------------------------------------------------------ per-env data */
typedef struct ref_data {
  /* persistent state */
  real *qpos, *qvel, *act, *ctrl, *qacc_warmstart;
  real time;
  /* position-dependent */
  real *xpos, *xquat, *xmat, *xipos, *ximat, *xanchor, *xaxis, *subtree_com;
  real *geom_xpos, *geom_xmat;
  real *cinert, *crb, *cdof, *cdof_dot, *cvel, *cacc, *cfrc;
  real *qM, *qLD, *qLDiagInv;
  real *con_dist, *con_pos, *con_frame;
  real *efc_J, *efc_D, *efc_aref, *efc_force, *efc_pos;
  real *qfrc_passive, *qfrc_bias, *qfrc_actuator, *qfrc_smooth, *qacc_smooth, *qfrc_constraint, *qacc;
  real *act_dot, *actuator_force;
  /* solver scratch */
  real *Jaref, *Ma, *grad, *Mgrad, *search, *mv, *jv, *tmpv, *quad;
  int solver_niter;
  real solver_cost;
  /* bookkeeping for the parity tests: which discrete decisions the trajectory took since the last ref_sig_reset */
  uint64_t sig_active;  /* FNV-1a over every forward pass's contact (dist < 0) and limit (pos < 0) bit patterns */
  uint64_t sig_rows;    /* same over the solver's final row states (Jaref < 0) */
  long niter_sum;       /* CG iterations summed over the forward passes */
  long nforward;
} ref_data;

static real* ralloc(size_t n) { return (real*)calloc(n ? n : 1, sizeof(real)); }

ref_data* ref_data_new(const ref_model* m) {
  ref_data* d = (ref_data*)calloc(1, sizeof(ref_data));
  int nb = m->nbody, nv = m->nv;
  d->qpos = ralloc(m->nq); d->qvel = ralloc(nv); d->act = ralloc(m->na); d->ctrl = ralloc(m->nu);
  d->qacc_warmstart = ralloc(nv);
  d->xpos = ralloc(nb * 3); d->xquat = ralloc(nb * 4); d->xmat = ralloc(nb * 9); d->xipos = ralloc(nb * 3);
  d->ximat = ralloc(nb * 9); d->xanchor = ralloc(m->njnt * 3); d->xaxis = ralloc(m->njnt * 3);
  d->subtree_com = ralloc(nb * 3); d->geom_xpos = ralloc(m->ngeom * 3); d->geom_xmat = ralloc(m->ngeom * 9);
  d->cinert = ralloc(nb * 10); d->crb = ralloc(nb * 10); d->cdof = ralloc(nv * 6); d->cdof_dot = ralloc(nv * 6);
  d->cvel = ralloc(nb * 6); d->cacc = ralloc(nb * 6); d->cfrc = ralloc(nb * 6);
  d->qM = ralloc(m->nM); d->qLD = ralloc(m->nM); d->qLDiagInv = ralloc(nv);
  d->con_dist = ralloc(m->ncon); d->con_pos = ralloc(m->ncon * 3); d->con_frame = ralloc(m->ncon * 9);
  d->efc_J = ralloc((size_t)m->nefc * nv); d->efc_D = ralloc(m->nefc); d->efc_aref = ralloc(m->nefc);
  d->efc_force = ralloc(m->nefc); d->efc_pos = ralloc(m->nefc);
  d->qfrc_passive = ralloc(nv); d->qfrc_bias = ralloc(nv); d->qfrc_actuator = ralloc(nv); d->qfrc_smooth = ralloc(nv);
  d->qacc_smooth = ralloc(nv); d->qfrc_constraint = ralloc(nv); d->qacc = ralloc(nv);
  d->act_dot = ralloc(m->na); d->actuator_force = ralloc(m->nu);
  d->Jaref = ralloc(m->nefc); d->Ma = ralloc(nv); d->grad = ralloc(nv); d->Mgrad = ralloc(nv); d->search = ralloc(nv);
  d->mv = ralloc(nv); d->jv = ralloc(m->nefc); d->tmpv = ralloc(nv); d->quad = ralloc(3 * (size_t)m->nefc);
  for (int i = 0; i < m->nq; i++) d->qpos[i] = m->qpos0[i];
  d->xquat[0] = 1; d->xmat[0] = d->xmat[4] = d->xmat[8] = 1;
  d->sig_active = d->sig_rows = 1469598103934665603ULL;
  return d;
}

void ref_data_free(ref_data* d) {
  real** ptrs[] = {&d->qpos, &d->qvel, &d->act, &d->ctrl, &d->qacc_warmstart, &d->xpos, &d->xquat, &d->xmat, &d->xipos,
                   &d->ximat, &d->xanchor, &d->xaxis, &d->subtree_com, &d->geom_xpos, &d->geom_xmat, &d->cinert, &d->crb,
                   &d->cdof, &d->cdof_dot, &d->cvel, &d->cacc, &d->cfrc, &d->qM, &d->qLD, &d->qLDiagInv, &d->con_dist,
                   &d->con_pos, &d->con_frame, &d->efc_J, &d->efc_D, &d->efc_aref, &d->efc_force, &d->efc_pos,
                   &d->qfrc_passive, &d->qfrc_bias, &d->qfrc_actuator, &d->qfrc_smooth, &d->qacc_smooth,
                   &d->qfrc_constraint, &d->qacc, &d->act_dot, &d->actuator_force, &d->Jaref, &d->Ma, &d->grad,
                   &d->Mgrad, &d->search, &d->mv, &d->jv, &d->tmpv, &d->quad};
  for (size_t i = 0; i < sizeof(ptrs) / sizeof(ptrs[0]); i++) free(*ptrs[i]);
  free(d);
}

/* ------------------------------------------------------------------ small math */
static void quat_mul(real* r, const real* a, const real* b) {
  real w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  real x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  real y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  real z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
static void quat_to_mat(real* m, const real* q) {
  real w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}
static void mat_vec(real* r, const real* m, const real* v) {
  real a = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  real b = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  real c = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = a; r[1] = b; r[2] = c;
}
static void quat_rot(real* r, const real* q, const real* v) { real m[9]; quat_to_mat(m, q); mat_vec(r, m, v); }
static void quat_normalize(real* q) {
  real n = SQRT(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  for (int i = 0; i < 4; i++) q[i] /= n;
}
static void cross(real* r, const real* a, const real* b) {
  real x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static real dot3(const real* a, const real* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static real dotn(const real* a, const real* b, int n) { real s = 0; for (int i = 0; i < n; i++) s += a[i] * b[i]; return s; }
static void axis_angle_quat(real* q, const real* axis, real angle) {
  real s = SIN(angle * R(0.5));
  q[0] = COS(angle * R(0.5)); q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
/* spatial inertia (10: xx yy zz xy xz yz, m*off(3), m) times motion vector (ang; lin)  [MuJoCo mju_mulInertVec] */
static void mul_inert_vec(real* res, const real* i, const real* v) {
  res[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  res[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  res[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  res[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  res[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  res[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}
static void cross_motion(real* res, const real* vel, const real* v) {
  res[0] = -vel[2] * v[1] + vel[1] * v[2];
  res[1] = vel[2] * v[0] - vel[0] * v[2];
  res[2] = -vel[1] * v[0] + vel[0] * v[1];
  res[3] = -vel[2] * v[4] + vel[1] * v[5] - vel[5] * v[1] + vel[4] * v[2];
  res[4] = vel[2] * v[3] - vel[0] * v[5] + vel[5] * v[0] - vel[3] * v[2];
  res[5] = -vel[1] * v[3] + vel[0] * v[4] - vel[4] * v[0] + vel[3] * v[1];
}
static void cross_force(real* res, const real* vel, const real* f) {
  res[0] = -vel[2] * f[1] + vel[1] * f[2] - vel[5] * f[4] + vel[4] * f[5];
  res[1] = vel[2] * f[0] - vel[0] * f[2] + vel[5] * f[3] - vel[3] * f[5];
  res[2] = -vel[1] * f[0] + vel[0] * f[1] - vel[4] * f[3] + vel[3] * f[4];
  res[3] = -vel[2] * f[4] + vel[1] * f[5];
  res[4] = vel[2] * f[3] - vel[0] * f[5];
  res[5] = -vel[1] * f[3] + vel[0] * f[4];
}

/* ------------------------------------------------------------------ A-1 kinematics [UP mjx smooth.kinematics] */
static void kinematics(const ref_model* m, ref_data* d) {
  for (int b = 1; b < m->nbody; b++) {
    int p = m->body_parentid[b];
    real pos[3], quat[4], t[3];
    mat_vec(t, d->xmat + 9 * p, m->body_pos + 3 * b);
    for (int k = 0; k < 3; k++) pos[k] = d->xpos[3 * p + k] + t[k];
    quat_mul(quat, d->xquat + 4 * p, m->body_quat + 4 * b);
    for (int jj = 0; jj < m->body_jntnum[b]; jj++) {
      int j = m->body_jntadr[b] + jj, qa = m->jnt_qposadr[j];
      if (m->jnt_type[j] == JNT_FREE) {
        for (int k = 0; k < 3; k++) pos[k] = d->qpos[qa + k];
        for (int k = 0; k < 4; k++) quat[k] = d->qpos[qa + 3 + k];
        quat_normalize(quat);
        for (int k = 0; k < 3; k++) { d->xanchor[3 * j + k] = pos[k]; d->xaxis[3 * j + k] = (k == 2); }
      } else { /* hinge */
        real anchor[3], vec[3], qloc[4], qn[4];
        quat_rot(anchor, quat, m->jnt_pos + 3 * j);
        for (int k = 0; k < 3; k++) anchor[k] += pos[k];
        quat_rot(d->xaxis + 3 * j, quat, m->jnt_axis + 3 * j);
        for (int k = 0; k < 3; k++) d->xanchor[3 * j + k] = anchor[k];
        axis_angle_quat(qloc, m->jnt_axis + 3 * j, d->qpos[qa] - m->qpos0[qa]);
        quat_mul(qn, quat, qloc);
        memcpy(quat, qn, sizeof(qn));
        quat_rot(vec, quat, m->jnt_pos + 3 * j);
        for (int k = 0; k < 3; k++) pos[k] = anchor[k] - vec[k];
      }
    }
    quat_normalize(quat);
    memcpy(d->xpos + 3 * b, pos, sizeof(pos));
    memcpy(d->xquat + 4 * b, quat, sizeof(quat));
    quat_to_mat(d->xmat + 9 * b, quat);
    mat_vec(t, d->xmat + 9 * b, m->body_ipos + 3 * b);
    for (int k = 0; k < 3; k++) d->xipos[3 * b + k] = pos[k] + t[k];
    real qi[4];
    quat_mul(qi, quat, m->body_iquat + 4 * b);
    quat_to_mat(d->ximat + 9 * b, qi);
  }
  for (int g = 0; g < m->ngeom; g++) {
    int b = m->geom_bodyid[g];
    real t[3], q[4];
    mat_vec(t, d->xmat + 9 * b, m->geom_pos + 3 * g);
    for (int k = 0; k < 3; k++) d->geom_xpos[3 * g + k] = d->xpos[3 * b + k] + t[k];
    quat_mul(q, d->xquat + 4 * b, m->geom_quat + 4 * g);
    quat_to_mat(d->geom_xmat + 9 * g, q);
  }
}

/* ------------------------------------------------------------------ A-2 com_pos [UP mjx smooth.com_pos] */
static void com_pos(const ref_model* m, ref_data* d) {
  int nb = m->nbody;
  real* mass_sub = (real*)calloc(nb, sizeof(real));
  for (int b = 0; b < nb; b++) {
    mass_sub[b] = m->body_mass[b];
    for (int k = 0; k < 3; k++) d->subtree_com[3 * b + k] = m->body_mass[b] * d->xipos[3 * b + k];
  }
  for (int b = nb - 1; b > 0; b--) {
    int p = m->body_parentid[b];
    mass_sub[p] += mass_sub[b];
    for (int k = 0; k < 3; k++) d->subtree_com[3 * p + k] += d->subtree_com[3 * b + k];
  }
  for (int b = 0; b < nb; b++)
    for (int k = 0; k < 3; k++)
      d->subtree_com[3 * b + k] = mass_sub[b] < MINVAL ? d->xipos[3 * b + k] : d->subtree_com[3 * b + k] / mass_sub[b];
  free(mass_sub);
  /* cinert: body inertia about the root's subtree COM, world axes [mju_inertCom] */
  for (int b = 1; b < nb; b++) {
    const real* R_ = d->ximat + 9 * b;
    const real* I = m->body_inertia + 3 * b;
    real mass = m->body_mass[b], dif[3];
    const real* com = d->subtree_com + 3 * m->body_rootid[b];
    for (int k = 0; k < 3; k++) dif[k] = d->xipos[3 * b + k] - com[k];
    real* c = d->cinert + 10 * b;
    /* R diag(I) R' */
    real t[9];
    for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) t[3 * r + k] = R_[3 * r + k] * I[k];
    c[0] = t[0] * R_[0] + t[1] * R_[1] + t[2] * R_[2];
    c[1] = t[3] * R_[3] + t[4] * R_[4] + t[5] * R_[5];
    c[2] = t[6] * R_[6] + t[7] * R_[7] + t[8] * R_[8];
    c[3] = t[0] * R_[3] + t[1] * R_[4] + t[2] * R_[5];
    c[4] = t[0] * R_[6] + t[1] * R_[7] + t[2] * R_[8];
    c[5] = t[3] * R_[6] + t[4] * R_[7] + t[5] * R_[8];
    c[0] += mass * (dif[1] * dif[1] + dif[2] * dif[2]);
    c[1] += mass * (dif[0] * dif[0] + dif[2] * dif[2]);
    c[2] += mass * (dif[0] * dif[0] + dif[1] * dif[1]);
    c[3] -= mass * dif[0] * dif[1];
    c[4] -= mass * dif[0] * dif[2];
    c[5] -= mass * dif[1] * dif[2];
    c[6] = mass * dif[0]; c[7] = mass * dif[1]; c[8] = mass * dif[2];
    c[9] = mass;
  }
  /* cdof [mju_dofCom] */
  for (int j = 0; j < m->njnt; j++) {
    int b = m->jnt_bodyid[j], da = m->jnt_dofadr[j];
    real off[3];
    for (int k = 0; k < 3; k++) off[k] = d->subtree_com[3 * m->body_rootid[b] + k] - d->xanchor[3 * j + k];
    if (m->jnt_type[j] == JNT_FREE) {
      for (int k = 0; k < 3; k++) {
        real* c = d->cdof + 6 * (da + k);
        for (int i = 0; i < 6; i++) c[i] = 0;
        c[3 + k] = 1;
      }
      for (int k = 0; k < 3; k++) {
        real* c = d->cdof + 6 * (da + 3 + k);
        real ax[3] = {d->xmat[9 * b + k], d->xmat[9 * b + 3 + k], d->xmat[9 * b + 6 + k]};
        c[0] = ax[0]; c[1] = ax[1]; c[2] = ax[2];
        cross(c + 3, ax, off);
      }
    } else {
      real* c = d->cdof + 6 * da;
      for (int k = 0; k < 3; k++) c[k] = d->xaxis[3 * j + k];
      cross(c + 3, d->xaxis + 3 * j, off);
    }
  }
}

/* ------------------------------------------------------------------ A-3 crb + factor [UP mjx smooth.crb/factor_m] */
static void crb(const ref_model* m, ref_data* d) {
  int nb = m->nbody, nv = m->nv;
  memcpy(d->crb, d->cinert, sizeof(real) * 10 * nb);
  for (int k = 0; k < 10; k++) d->crb[k] = 0;
  for (int b = nb - 1; b > 0; b--) {
    int p = m->body_parentid[b];
    if (p > 0) for (int k = 0; k < 10; k++) d->crb[10 * p + k] += d->crb[10 * b + k];
  }
  for (int i = 0; i < nv; i++) {
    real buf[6];
    mul_inert_vec(buf, d->crb + 10 * m->dof_bodyid[i], d->cdof + 6 * i);
    int adr = m->dof_Madr[i];
    d->qM[adr] = m->dof_armature[i];
    int j = i, k = 0;
    while (j >= 0) {
      real v = dotn(d->cdof + 6 * j, buf, 6);
      if (k == 0) d->qM[adr] += v; else d->qM[adr + k] = v;
      j = m->dof_parentid[j];
      k++;
    }
  }
}

/* sparse L'DL of a tree-structured matrix in MuJoCo's row layout [MuJoCo mj_factorM] */
static void factor_m(const ref_model* m, const real* qM, real* qLD, real* diaginv) {
  int nv = m->nv;
  memcpy(qLD, qM, sizeof(real) * m->nM);
  for (int k = nv - 1; k >= 0; k--) {
    int Mkk = m->dof_Madr[k];
    int i = m->dof_parentid[k], Mki = Mkk + 1;
    while (i >= 0) {
      real tmp = qLD[Mki] / qLD[Mkk];
      int Mii = m->dof_Madr[i], cnt = m->dof_depth[i] + 1;
      for (int c = 0; c < cnt; c++) qLD[Mii + c] -= qLD[Mki + c] * tmp;
      qLD[Mki] = tmp;
      i = m->dof_parentid[i];
      Mki++;
    }
    diaginv[k] = 1 / qLD[Mkk];
  }
}
static void solve_ld(const ref_model* m, const real* qLD, const real* diaginv, real* x) {
  int nv = m->nv;
  for (int i = nv - 1; i >= 0; i--) {
    int a = m->dof_Madr[i] + 1, j = m->dof_parentid[i];
    while (j >= 0) { x[j] -= qLD[a++] * x[i]; j = m->dof_parentid[j]; }
  }
  for (int i = 0; i < nv; i++) x[i] *= diaginv[i];
  for (int i = 0; i < nv; i++) {
    int a = m->dof_Madr[i] + 1, j = m->dof_parentid[i];
    while (j >= 0) { x[i] -= qLD[a++] * x[j]; j = m->dof_parentid[j]; }
  }
}
static void mul_m(const ref_model* m, const real* qM, real* res, const real* v) {
  int nv = m->nv;
  for (int i = 0; i < nv; i++) res[i] = 0;
  for (int i = 0; i < nv; i++) {
    int a = m->dof_Madr[i];
    res[i] += qM[a] * v[i];
    int j = m->dof_parentid[i];
    a++;
    while (j >= 0) {
      res[i] += qM[a] * v[j];
      res[j] += qM[a] * v[i];
      j = m->dof_parentid[j];
      a++;
    }
  }
}

/* ------------------------------------------------------------------ A-4 collision [UP mjx collision_driver / collision_primitive] */
/* closest point of segment [a, b] to pt  [UP mjx math.closest_segment_point: t = (pt - a).ab / (ab.ab + 1e-6), clipped to [0, 1]] */
static void closest_segment_point(real* out, const real* a, const real* b, const real* pt) {
  real ab[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, pa[3] = {pt[0] - a[0], pt[1] - a[1], pt[2] - a[2]};
  real t = dot3(pa, ab) / (dot3(ab, ab) + R(1e-6));
  t = FMIN(FMAX(t, 0), 1);
  for (int k = 0; k < 3; k++) out[k] = a[k] + t * ab[k];
}
/* closest points of two segments  [UP mjx math.closest_segment_to_segment_points]: closest points of the two LINES (parametrised from the
 * segment mid-points, denominator + 1e-6), clipped to the segments; then the case where both were clipped is resolved by re-projecting
 * each clipped point on the other segment and keeping the closer pair. */
static void closest_segment_to_segment(real* best_a, real* best_b, const real* a0, const real* a1, const real* b0, const real* b1) {
  real da[3], db[3], amid[3], bmid[3], trans[3];
  real la = 0, lb = 0;
  for (int k = 0; k < 3; k++) { da[k] = a1[k] - a0[k]; db[k] = b1[k] - b0[k]; la += da[k] * da[k]; lb += db[k] * db[k]; }
  la = SQRT(la); lb = SQRT(lb);
  for (int k = 0; k < 3; k++) { da[k] = la > 0 ? da[k] / la : 0; db[k] = lb > 0 ? db[k] / lb : 0; }
  real ha = la * R(0.5), hb = lb * R(0.5);
  for (int k = 0; k < 3; k++) { amid[k] = a0[k] + da[k] * ha; bmid[k] = b0[k] + db[k] * hb; trans[k] = amid[k] - bmid[k]; }
  real dab = dot3(da, db), dat = dot3(da, trans), dbt = dot3(db, trans);
  real denom = 1 - dab * dab;
  real ta = (-dat + dab * dbt) / (denom + R(1e-6));
  real tb = dbt + ta * dab;
  ta = FMIN(FMAX(ta, -ha), ha);
  tb = FMIN(FMAX(tb, -hb), hb);
  for (int k = 0; k < 3; k++) { best_a[k] = amid[k] + da[k] * ta; best_b[k] = bmid[k] + db[k] * tb; }
  real na[3], nb_[3];
  closest_segment_point(na, a0, a1, best_b);
  closest_segment_point(nb_, b0, b1, best_a);
  real d1 = 0, d2 = 0;
  for (int k = 0; k < 3; k++) { d1 += (na[k] - best_b[k]) * (na[k] - best_b[k]); d2 += (nb_[k] - best_a[k]) * (nb_[k] - best_a[k]); }
  if (d1 < d2) { for (int k = 0; k < 3; k++) best_a[k] = na[k]; }
  else { for (int k = 0; k < 3; k++) best_b[k] = nb_[k]; }
}
/* make_frame(n) [UP mjx math.make_frame]: rows n, b = normalise(y or z minus its part along n), n x b */
static void make_frame(real* fr, const real* n) {
  real b[3] = {0, 0, 0};
  if (n[1] > R(-0.5) && n[1] < R(0.5)) b[1] = 1; else b[2] = 1;
  real nb_ = dot3(n, b);
  for (int k = 0; k < 3; k++) b[k] -= n[k] * nb_;
  real bn = SQRT(dot3(b, b));
  for (int k = 0; k < 3; k++) { b[k] /= bn; fr[k] = n[k]; fr[3 + k] = b[k]; }
  cross(fr + 6, n, b);
}
/* two spheres (also the end of sphere-capsule / capsule-capsule)  [UP mjx collision_primitive._sphere_sphere]: normal from 1 to 2
 * ((1, 0, 0) when the centres coincide), dist = |p2 - p1| - r1 - r2, pos = p1 + n (r1 + dist / 2) */
static void sphere_sphere(real* dist_, real* pos, real* fr, const real* p1, real r1, const real* p2, real r2) {
  real n[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  real len = SQRT(dot3(n, n));
  if (len == 0) { n[0] = 1; n[1] = n[2] = 0; } else { for (int k = 0; k < 3; k++) n[k] /= len; }
  real dist = len - (r1 + r2);
  for (int k = 0; k < 3; k++) pos[k] = p1[k] + n[k] * (r1 + dist * R(0.5));
  make_frame(fr, n);
  *dist_ = dist;
}

static void collision(const ref_model* m, ref_data* d) {
  for (int c = 0; c < m->ncon; c++) {
    int g1 = m->con_geom1[c], g2 = m->con_geom2[c], kind = m->con_kind[c];
    if (kind >= CON_SPHERE_SPHERE) {     /* two moving geoms: sphere-sphere, sphere-capsule, capsule-capsule [UP mjx collision_primitive] */
      const real *m1 = d->geom_xmat + 9 * g1, *p1 = d->geom_xpos + 3 * g1, *s1 = m->geom_size + 3 * g1;
      const real *m2 = d->geom_xmat + 9 * g2, *p2 = d->geom_xpos + 3 * g2, *s2 = m->geom_size + 3 * g2;
      real a[3] = {p1[0], p1[1], p1[2]}, b[3] = {p2[0], p2[1], p2[2]};
      real e20[3], e21[3];
      for (int k = 0; k < 3; k++) { e20[k] = p2[k] - m2[3 * k + 2] * s2[1]; e21[k] = p2[k] + m2[3 * k + 2] * s2[1]; }
      if (kind == CON_SPHERE_CAPSULE) {
        closest_segment_point(b, e20, e21, p1);
      } else if (kind == CON_CAPSULE_CAPSULE) {
        real e10[3], e11[3];
        for (int k = 0; k < 3; k++) { e10[k] = p1[k] - m1[3 * k + 2] * s1[1]; e11[k] = p1[k] + m1[3 * k + 2] * s1[1]; }
        closest_segment_to_segment(a, b, e10, e11, e20, e21);
      }
      sphere_sphere(d->con_dist + c, d->con_pos + 3 * c, d->con_frame + 9 * c, a, s1[0], b, s2[0]);
      continue;
    }
    const real* pm = d->geom_xmat + 9 * g1;
    const real* pp = d->geom_xpos + 3 * g1;
    const real* gm = d->geom_xmat + 9 * g2;
    const real* gp = d->geom_xpos + 3 * g2;
    const real* size = m->geom_size + 3 * g2;
    real n[3] = {pm[2], pm[5], pm[8]};
    real* fr = d->con_frame + 9 * c;
    real* pos = d->con_pos + 3 * c;
    if (kind == CON_ELLIPSOID) {
      /* support point of the ellipsoid along -n */
      real s[3], nn = 0;
      for (int k = 0; k < 3; k++) { s[k] = (gm[k] * n[0] + gm[3 + k] * n[1] + gm[6 + k] * n[2]) * size[k]; nn += s[k] * s[k]; }
      nn = SQRT(nn);
      for (int k = 0; k < 3; k++) s[k] = (nn < MINVAL ? 0 : -s[k] / nn) * size[k];
      real p[3];
      mat_vec(p, gm, s);
      real dist = 0;
      for (int k = 0; k < 3; k++) { p[k] += gp[k]; dist += n[k] * (p[k] - pp[k]); }
      for (int k = 0; k < 3; k++) pos[k] = p[k] - n[k] * dist * R(0.5);
      d->con_dist[c] = dist;
      /* make_frame(n) */
      real b[3] = {0, 0, 0};
      if (n[1] > R(-0.5) && n[1] < R(0.5)) b[1] = 1; else b[2] = 1;
      real nb_ = dot3(n, b);
      for (int k = 0; k < 3; k++) b[k] -= n[k] * nb_;
      real bn = SQRT(dot3(b, b));
      for (int k = 0; k < 3; k++) b[k] /= bn;
      for (int k = 0; k < 3; k++) { fr[k] = n[k]; fr[3 + k] = b[k]; }
      cross(fr + 6, n, b);
    } else {
      real center[3] = {gp[0], gp[1], gp[2]};
      real radius = size[0];
      if (kind == CON_SPHERE) {
        real b[3] = {0, 0, 0};
        if (n[1] > R(-0.5) && n[1] < R(0.5)) b[1] = 1; else b[2] = 1;
        real nb_ = dot3(n, b);
        for (int k = 0; k < 3; k++) b[k] -= n[k] * nb_;
        real bn = SQRT(dot3(b, b));
        for (int k = 0; k < 3; k++) { b[k] /= bn; fr[k] = n[k]; fr[3 + k] = b[k]; }
        cross(fr + 6, n, b);
      } else {
        real axis[3] = {gm[2], gm[5], gm[8]};
        real na = dot3(n, axis), b[3];
        for (int k = 0; k < 3; k++) b[k] = axis[k] - n[k] * na;
        real bn = SQRT(dot3(b, b));
        if (bn < R(0.5)) {
          b[0] = b[1] = b[2] = 0;
          if (n[1] > R(-0.5) && n[1] < R(0.5)) b[1] = 1; else b[2] = 1;
        } else {
          for (int k = 0; k < 3; k++) b[k] /= bn;
        }
        for (int k = 0; k < 3; k++) { fr[k] = n[k]; fr[3 + k] = b[k]; }
        cross(fr + 6, n, b);
        real sgn = kind == CON_CAP_POS ? R(1.0) : R(-1.0);
        for (int k = 0; k < 3; k++) center[k] += sgn * axis[k] * size[1];
      }
      real dist = -radius;
      for (int k = 0; k < 3; k++) dist += (center[k] - pp[k]) * n[k];
      for (int k = 0; k < 3; k++) pos[k] = center[k] - n[k] * (radius + R(0.5) * dist);
      d->con_dist[c] = dist;
    }
  }
}

/* ------------------------------------------------------------------ A-5 make_constraint [UP mjx constraint.make_constraint] */
static void kbi(const ref_model* m, const real* solref, const real* solimp, real pos, real* k_, real* b_, real* imp_) {
  real timeconst = FMAX(solref[0], 2 * m->timestep); /* refsafe */
  real dampratio = solref[1];
  real dmin = FMIN(FMAX(solimp[0], MINIMP), MAXIMP), dmax = FMIN(FMAX(solimp[1], MINIMP), MAXIMP);
  real width = FMAX(MINVAL, solimp[2]);
  real mid = FMIN(FMAX(solimp[3], MINIMP), MAXIMP);
  real power = FMAX(R(1.0), solimp[4]);
  real k = 1 / (dmax * dmax * timeconst * timeconst * dampratio * dampratio);
#ifdef REF_BUG   /* deliberate modelling error for tests/test_parity_criteria.py: 5 % too much constraint damping */
  real b = R(2.1) / (dmax * timeconst);
#else
  real b = 2 / (dmax * timeconst);
#endif
  if (solref[0] <= 0) k = -solref[0] / (dmax * dmax);
  if (solref[1] <= 0) b = -solref[1] / dmax;
  real x = FABS(pos) / width;
  real a_ = (1 / POW(mid, power - 1)) * POW(x, power);
  real bb = 1 - (1 / POW(1 - mid, power - 1)) * POW(1 - x, power);
  real y = x < mid ? a_ : bb;
  real imp = dmin + y * (dmax - dmin);
  imp = FMIN(FMAX(imp, dmin), dmax);
  if (x > 1) imp = dmax;
  *k_ = k; *b_ = b; *imp_ = imp;
}

static void make_constraint(const ref_model* m, ref_data* d) {
  int nv = m->nv;
  memset(d->efc_J, 0, sizeof(real) * (size_t)m->nefc * nv);
  int row = 0;
  for (int l = 0; l < m->nlimit; l++, row++) {
    int j = m->limit_jnt[l], qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    real q = d->qpos[qa];
    real dmin_ = q - m->jnt_range[2 * j], dmax_ = m->jnt_range[2 * j + 1] - q;
    real pos = FMIN(dmin_, dmax_);
    int active = pos < 0;
    real k, b, imp;
    real* J = d->efc_J + (size_t)row * nv;
    J[da] = active ? (dmin_ < dmax_ ? R(1.0) : R(-1.0)) : 0;
    kbi(m, m->jnt_solref + 2 * j, m->jnt_solimp + 5 * j, pos, &k, &b, &imp);
    real r = FMAX(m->dof_invweight0[da] * (1 - imp) / imp, MINVAL);
    d->efc_D[row] = 1 / r;
    d->efc_pos[row] = pos;
    d->efc_aref[row] = -b * (J[da] * d->qvel[da]) - k * imp * pos;
  }
  for (int c = 0; c < m->ncon; c++) {
    int body = m->con_body2[c];
    real dist = d->con_dist[c];
    int active = dist < 0;
    const real* fr = d->con_frame + 9 * c;
    const real* com = d->subtree_com + 3 * m->body_rootid[body];
    real off[3];
    for (int k = 0; k < 3; k++) off[k] = d->con_pos[3 * c + k] - com[k];
    real mu = m->con_friction[5 * c];
    real k, b, imp;
    kbi(m, m->con_solref + 2 * c, m->con_solimp + 5 * c, dist, &k, &b, &imp);
    const int frictionless = m->con_dim[c] == 1;      /* condim 1: ONE row along the normal, invweight = the two bodies' [UP mjx constraint._instantiate_contact] */
    real invw = frictionless ? m->con_invweight[c] : (m->con_invweight[c] + mu * mu * m->con_invweight[c]) * 2 * mu * mu / m->impratio;
    real r = FMAX(invw * (1 - imp) / imp, MINVAL);
    real* J0 = d->efc_J + (size_t)row * nv;
    if (active) {      /* J = jac(body2, pos) - jac(body1, pos) projected on the frame; a world body1 (floor contacts) has no dofs */
      for (int side = 0; side < 2; side++) {
        const real sg = side == 0 ? R(1.0) : R(-1.0);
        int dd = m->body_lastdof[side == 0 ? body : m->con_body1[c]];
        while (dd >= 0) {
          const real* cd = d->cdof + 6 * dd;
          real jp[3], t[3];
          cross(t, cd, off);
          for (int kk = 0; kk < 3; kk++) jp[kk] = cd[3 + kk] + t[kk];
          real jn = dot3(fr, jp), j1 = dot3(fr + 3, jp), j2 = dot3(fr + 6, jp);
          if (frictionless) {
            J0[dd] += sg * jn;
          } else {
            J0[dd] += sg * (jn + mu * j1);
            J0[nv + dd] += sg * (jn - mu * j1);
            J0[2 * nv + dd] += sg * (jn + mu * j2);
            J0[3 * nv + dd] += sg * (jn - mu * j2);
          }
          dd = m->dof_parentid[dd];
        }
      }
    }
    for (int kk = 0; kk < (frictionless ? 1 : 4); kk++, row++) {
      d->efc_D[row] = 1 / r;
      d->efc_pos[row] = dist;
      d->efc_aref[row] = -b * dotn(d->efc_J + (size_t)row * nv, d->qvel, nv) - k * imp * dist;
    }
  }
}

/* ------------------------------------------------------------------ A-6 velocity, passive, rne, actuation */
static void com_vel(const ref_model* m, ref_data* d) {
  for (int k = 0; k < 6; k++) d->cvel[k] = 0;
  for (int b = 1; b < m->nbody; b++) {
    real v[6];
    memcpy(v, d->cvel + 6 * m->body_parentid[b], sizeof(v));
    for (int jj = 0; jj < m->body_jntnum[b]; jj++) {
      int j = m->body_jntadr[b] + jj, da = m->jnt_dofadr[j];
      if (m->jnt_type[j] == JNT_FREE) {
        for (int k = 0; k < 18; k++) d->cdof_dot[6 * da + k] = 0;
        for (int k = 0; k < 3; k++) for (int i = 0; i < 6; i++) v[i] += d->cdof[6 * (da + k) + i] * d->qvel[da + k];
        for (int k = 3; k < 6; k++) cross_motion(d->cdof_dot + 6 * (da + k), v, d->cdof + 6 * (da + k));
        for (int k = 3; k < 6; k++) for (int i = 0; i < 6; i++) v[i] += d->cdof[6 * (da + k) + i] * d->qvel[da + k];
      } else {
        cross_motion(d->cdof_dot + 6 * da, v, d->cdof + 6 * da);
        for (int i = 0; i < 6; i++) v[i] += d->cdof[6 * da + i] * d->qvel[da];
      }
    }
    memcpy(d->cvel + 6 * b, v, sizeof(v));
  }
}
static void passive(const ref_model* m, ref_data* d) {
  for (int i = 0; i < m->nv; i++) d->qfrc_passive[i] = -m->dof_damping[i] * d->qvel[i];
  for (int j = 0; j < m->njnt; j++)
    if (m->jnt_type[j] == JNT_HINGE) {
      int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
      d->qfrc_passive[da] -= m->jnt_stiffness[j] * (d->qpos[qa] - m->qpos_spring[qa]);
    }
}
static void rne(const ref_model* m, ref_data* d) {
  int nb = m->nbody;
  for (int k = 0; k < 3; k++) { d->cacc[k] = 0; d->cacc[3 + k] = -m->gravity[k]; }
  for (int k = 0; k < 6; k++) d->cfrc[k] = 0;
  for (int b = 1; b < nb; b++) {
    real* a = d->cacc + 6 * b;
    memcpy(a, d->cacc + 6 * m->body_parentid[b], 6 * sizeof(real));
    for (int k = 0; k < m->body_dofnum[b]; k++) {
      int dd = m->body_dofadr[b] + k;
      for (int i = 0; i < 6; i++) a[i] += d->cdof_dot[6 * dd + i] * d->qvel[dd];
    }
    real t[6], t1[6], t2[6];
    mul_inert_vec(t, d->cinert + 10 * b, d->cvel + 6 * b);
    cross_force(t1, d->cvel + 6 * b, t);
    mul_inert_vec(t2, d->cinert + 10 * b, a);
    for (int i = 0; i < 6; i++) d->cfrc[6 * b + i] = t2[i] + t1[i];
  }
  for (int b = nb - 1; b > 0; b--) {
    int p = m->body_parentid[b];
    if (p > 0) for (int i = 0; i < 6; i++) d->cfrc[6 * p + i] += d->cfrc[6 * b + i];
  }
  for (int i = 0; i < m->nv; i++) d->qfrc_bias[i] = dotn(d->cdof + 6 * i, d->cfrc + 6 * m->dof_bodyid[i], 6);
}
static void fwd_actuation(const ref_model* m, ref_data* d) {
  for (int i = 0; i < m->nv; i++) d->qfrc_actuator[i] = 0;
  for (int u = 0; u < m->nu; u++) {
    real c = FMIN(FMAX(d->ctrl[u], m->ctrlrange[2 * u]), m->ctrlrange[2 * u + 1]);
    d->act_dot[u] = (c - d->act[u]) / FMAX(m->tau[u], MINVAL);
    /* transmission [UP mjx smooth.transmission]: length = sum coef * qpos, velocity = sum coef * qvel (joint: one entry, gear 1;
     * fixed tendon [REF models/rodent_cpu.xml:505-560]: its joints), force mapped back through the same coefficients */
    real len = 0, vel = 0;
    for (int e = m->actuator_momentadr[u]; e < m->actuator_momentadr[u + 1]; e++) {
      len += m->actuator_moment_coef[e] * d->qpos[m->actuator_moment_qposadr[e]];
      vel += m->actuator_moment_coef[e] * d->qvel[m->actuator_moment_dofadr[e]];
    }
    real f = m->gain0[u] * d->act[u] + m->biasprm[3 * u] + m->biasprm[3 * u + 1] * len + m->biasprm[3 * u + 2] * vel;
    d->actuator_force[u] = f;
    for (int e = m->actuator_momentadr[u]; e < m->actuator_momentadr[u + 1]; e++)
      d->qfrc_actuator[m->actuator_moment_dofadr[e]] += m->actuator_moment_coef[e] * f;
  }
}
static void fwd_acceleration(const ref_model* m, ref_data* d) {
  for (int i = 0; i < m->nv; i++) {
    d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_actuator[i];
    d->qacc_smooth[i] = d->qfrc_smooth[i];
  }
  solve_ld(m, d->qLD, d->qLDiagInv, d->qacc_smooth);
}

/* ------------------------------------------------------------------ A-7 solver (primal CG) [UP mjx solver.solve] */
typedef struct { real alpha, cost, deriv0, deriv1; } ls_point;

typedef struct {
  real gauss, cost, prev_cost;
} ctx_scalars;

static void update_constraint(const ref_model* m, ref_data* d, ctx_scalars* s) {
  int nv = m->nv, nefc = m->nefc;
  real cost = 0;
  for (int i = 0; i < nv; i++) d->qfrc_constraint[i] = 0;
  for (int r = 0; r < nefc; r++) {
    int active = d->Jaref[r] < 0;
    real f = active ? d->efc_D[r] * -d->Jaref[r] : 0;
    d->efc_force[r] = f;
    if (active) {
      cost += d->efc_D[r] * d->Jaref[r] * d->Jaref[r];
      const real* J = d->efc_J + (size_t)r * nv;
      for (int i = 0; i < nv; i++) d->qfrc_constraint[i] += J[i] * f;
    }
  }
  real gauss = 0;
  for (int i = 0; i < nv; i++) gauss += (d->Ma[i] - d->qfrc_smooth[i]) * (d->qacc[i] - d->qacc_smooth[i]);
  gauss *= R(0.5);
  s->prev_cost = s->cost;
  s->gauss = gauss;
  s->cost = R(0.5) * cost + gauss;
}
/* Newton direction [UP mjx solver._update_gradient, SolverType.NEWTON; REF Rodent_Env_Brax.py:42-45 accepts solver='newton']:
 * H = M + J' diag(D * active) J (dense), Cholesky, Mgrad = H^-1 grad */
static void newton_direction(const ref_model* m, ref_data* d) {
  int nv = m->nv;
  real* H = (real*)calloc((size_t)nv * nv, sizeof(real));
  for (int i = 0; i < nv; i++) {                          /* full_m: the sparse mass matrix as a dense symmetric one */
    int a = m->dof_Madr[i], j = i;
    while (j >= 0) { H[i * nv + j] = H[j * nv + i] = d->qM[a++]; j = m->dof_parentid[j]; }
  }
  for (int r = 0; r < m->nefc; r++) {
    if (!(d->Jaref[r] < 0)) continue;
    const real* J = d->efc_J + (size_t)r * nv;
    for (int i = 0; i < nv; i++) {
      if (J[i] == 0) continue;
      real di = d->efc_D[r] * J[i];
      for (int j = 0; j < nv; j++) H[i * nv + j] += di * J[j];
    }
  }
  for (int k = 0; k < nv; k++) {                          /* H = L L', lower triangle in place */
    real p = SQRT(H[k * nv + k]);
    H[k * nv + k] = p;
    for (int i = k + 1; i < nv; i++) H[i * nv + k] /= p;
    for (int i = k + 1; i < nv; i++)
      for (int j = k + 1; j <= i; j++) H[i * nv + j] -= H[i * nv + k] * H[j * nv + k];
  }
  real* x = d->Mgrad;
  for (int i = 0; i < nv; i++) {
    real s = d->grad[i];
    for (int j = 0; j < i; j++) s -= H[i * nv + j] * x[j];
    x[i] = s / H[i * nv + i];
  }
  for (int i = nv - 1; i >= 0; i--) {
    real s = x[i];
    for (int j = i + 1; j < nv; j++) s -= H[j * nv + i] * x[j];
    x[i] = s / H[i * nv + i];
  }
  free(H);
}
static void update_gradient(const ref_model* m, ref_data* d) {
  for (int i = 0; i < m->nv; i++) {
    d->grad[i] = d->Ma[i] - d->qfrc_smooth[i] - d->qfrc_constraint[i];
    d->Mgrad[i] = d->grad[i];
  }
  if (m->solver == 2) newton_direction(m, d);
  else solve_ld(m, d->qLD, d->qLDiagInv, d->Mgrad);
}
static void ctx_create(const ref_model* m, ref_data* d, ctx_scalars* s, int grad) {
  int nv = m->nv;
  for (int r = 0; r < m->nefc; r++) d->Jaref[r] = dotn(d->efc_J + (size_t)r * nv, d->qacc, nv) - d->efc_aref[r];
  mul_m(m, d->qM, d->Ma, d->qacc);
  s->cost = INFINITY; s->prev_cost = 0; s->gauss = 0;
  update_constraint(m, d, s);
  if (grad) {
    update_gradient(m, d);
    for (int i = 0; i < nv; i++) d->search[i] = -d->Mgrad[i];
  }
}
static ls_point ls_eval(const ref_model* m, const ref_data* d, real alpha, const real* quad_gauss) {
  real q0 = quad_gauss[0], q1 = quad_gauss[1], q2 = quad_gauss[2];
  for (int r = 0; r < m->nefc; r++) {
    real x = d->Jaref[r] + alpha * d->jv[r];
    if (x < 0) { q0 += d->quad[3 * r]; q1 += d->quad[3 * r + 1]; q2 += d->quad[3 * r + 2]; }
  }
  ls_point p;
  p.alpha = alpha;
  p.cost = alpha * alpha * q2 + alpha * q1 + q0;
  p.deriv0 = 2 * alpha * q2 + q1;
  p.deriv1 = 2 * q2 + (q2 == 0 ? MINVAL : 0);
  return p;
}
static void linesearch(const ref_model* m, ref_data* d, ctx_scalars* s) {
  int nv = m->nv, nefc = m->nefc;
  real smag = SQRT(dotn(d->search, d->search, nv)) * m->meaninertia * (real)(nv > 1 ? nv : 1);
  real gtol = m->tolerance * m->ls_tolerance * smag;
  mul_m(m, d->qM, d->mv, d->search);
  for (int r = 0; r < nefc; r++) d->jv[r] = dotn(d->efc_J + (size_t)r * nv, d->search, nv);
  real quad_gauss[3] = {s->gauss, dotn(d->search, d->Ma, nv) - dotn(d->search, d->qfrc_smooth, nv),
                        R(0.5) * dotn(d->search, d->mv, nv)};
  for (int r = 0; r < nefc; r++) {
    d->quad[3 * r] = R(0.5) * d->Jaref[r] * d->Jaref[r] * d->efc_D[r];
    d->quad[3 * r + 1] = d->jv[r] * d->Jaref[r] * d->efc_D[r];
    d->quad[3 * r + 2] = R(0.5) * d->jv[r] * d->jv[r] * d->efc_D[r];
  }
  ls_point p0 = ls_eval(m, d, 0, quad_gauss);
  ls_point lo = ls_eval(m, d, p0.alpha - p0.deriv0 / p0.deriv1, quad_gauss);
  ls_point hi;
  if (lo.deriv0 < p0.deriv0) { hi = p0; } else { hi = lo; lo = p0; }
  /* note: lesser_fn(x,y)=where(lo.deriv_0<p0.deriv_0,x,y): hi=lesser(p0,lo), lo=lesser(lo,p0) */
  int swap = 1, it = 0;
  while (1) {
    int done = it >= m->ls_iterations;
    done |= !swap;
    done |= (lo.deriv0 < 0) && (lo.deriv0 > -gtol);
    done |= (hi.deriv0 > 0) && (hi.deriv0 < gtol);
    if (done) break;
    ls_point lo_next = ls_eval(m, d, lo.alpha - lo.deriv0 / lo.deriv1, quad_gauss);
    ls_point hi_next = ls_eval(m, d, hi.alpha - hi.deriv0 / hi.deriv1, quad_gauss);
    ls_point mid = ls_eval(m, d, R(0.5) * (lo.alpha + hi.alpha), quad_gauss);
    int swap_lo_next = (lo.deriv0 > 0) || (lo.deriv0 < lo_next.deriv0);
    if (swap_lo_next) lo = lo_next;
    int swap_lo_mid = (mid.deriv0 < 0) && (lo.deriv0 < mid.deriv0);
    if (swap_lo_mid) lo = mid;
    int swap_hi_next = (hi.deriv0 < 0) || (hi.deriv0 > hi_next.deriv0);
    if (swap_hi_next) hi = hi_next;
    int swap_hi_mid = (mid.deriv0 > 0) && (hi.deriv0 > mid.deriv0);
    if (swap_hi_mid) hi = mid;
    swap = swap_lo_next || swap_lo_mid || swap_hi_next || swap_hi_mid;
    it++;
  }
  int improved = (lo.cost < p0.cost) || (hi.cost < p0.cost);
  real alpha = lo.cost < hi.cost ? lo.alpha : hi.alpha;
  if (improved) {
    for (int i = 0; i < nv; i++) { d->qacc[i] += d->search[i] * alpha; d->Ma[i] += d->mv[i] * alpha; }
    for (int r = 0; r < nefc; r++) d->Jaref[r] += d->jv[r] * alpha;
  }
}
static void solve(const ref_model* m, ref_data* d) {
  int nv = m->nv;
  ctx_scalars s;
  real scale = 1 / (m->meaninertia * (real)(nv > 1 ? nv : 1));
  /* warm start: the cheaper of qacc_warmstart and qacc_smooth */
  memcpy(d->qacc, d->qacc_smooth, sizeof(real) * nv);
  ctx_create(m, d, &s, 0);
  real cost_smooth = s.cost;
  memcpy(d->qacc, d->qacc_warmstart, sizeof(real) * nv);
  ctx_create(m, d, &s, 0);
  if (!(s.cost < cost_smooth)) memcpy(d->qacc, d->qacc_smooth, sizeof(real) * nv);
  ctx_create(m, d, &s, 1);
  int niter = 0;
  while (1) {
    real improvement = (s.prev_cost - s.cost) * scale;
    real gradient = SQRT(dotn(d->grad, d->grad, nv)) * scale;
    int done = niter >= m->iterations;
    done |= improvement < m->tolerance;
    done |= gradient < m->tolerance;
    if (done) break;
    linesearch(m, d, &s);
    real* prev_grad = d->tmpv;
    real gg_prev = dotn(d->grad, d->Mgrad, nv);
    real* prev_Mgrad = d->mv; /* mv is dead after the line search */
    memcpy(prev_grad, d->grad, sizeof(real) * nv);
    memcpy(prev_Mgrad, d->Mgrad, sizeof(real) * nv);
    update_constraint(m, d, &s);
    update_gradient(m, d);
    real beta = 0;
    for (int i = 0; i < nv; i++) beta += d->grad[i] * (d->Mgrad[i] - prev_Mgrad[i]);
    beta = beta / FMAX(MINVAL, gg_prev);
    beta = FMAX(0, beta);
    if (m->solver == 2) beta = 0;            /* Newton: search = -Mgrad */
    for (int i = 0; i < nv; i++) d->search[i] = -d->Mgrad[i] + beta * d->search[i];
    niter++;
  }
  d->solver_niter = niter;
  d->solver_cost = s.cost;
  memcpy(d->qacc_warmstart, d->qacc, sizeof(real) * nv);
}

/* ------------------------------------------------------------------ forward / euler / step */
void ref_forward(const ref_model* m, ref_data* d) {
  kinematics(m, d);
  com_pos(m, d);
  crb(m, d);
  factor_m(m, d->qM, d->qLD, d->qLDiagInv);
  collision(m, d);
  make_constraint(m, d);
  com_vel(m, d);
  passive(m, d);
  rne(m, d);
  fwd_actuation(m, d);
  fwd_acceleration(m, d);
  solve(m, d);
  for (int r = 0; r < m->nefc; r++) {
    d->sig_active = (d->sig_active ^ (uint64_t)(d->efc_pos[r] < 0)) * 1099511628211ULL;
    d->sig_rows = (d->sig_rows ^ (uint64_t)(d->Jaref[r] < 0)) * 1099511628211ULL;
  }
  d->niter_sum += d->solver_niter;
  d->nforward += 1;
}
void ref_sig_reset(ref_data* d) { d->sig_active = d->sig_rows = 1469598103934665603ULL; d->niter_sum = 0; d->nforward = 0; }
void ref_sig_get(const ref_data* d, uint64_t* out /*[4]*/) { out[0] = d->sig_active; out[1] = d->sig_rows; out[2] = (uint64_t)d->niter_sum; out[3] = (uint64_t)d->nforward; }

/* A-8 [UP mjx forward.euler/_advance] */
static void euler(const ref_model* m, ref_data* d) {
  int nv = m->nv;
  real dt = m->timestep;
  real* qacc = d->tmpv;
  /* eulerdamp: (M + dt*diag(damping))^-1 (qfrc_smooth + qfrc_constraint) */
  real* qH = d->quad; /* nM <= 3*nefc not guaranteed: allocate */
  real* Hld = (real*)malloc(sizeof(real) * m->nM * 2 + sizeof(real) * nv);
  real* Hm = Hld + m->nM;
  real* Hdi = Hm + m->nM;
  (void)qH;
  memcpy(Hm, d->qM, sizeof(real) * m->nM);
  for (int i = 0; i < nv; i++) Hm[m->dof_Madr[i]] += dt * m->dof_damping[i];
  factor_m(m, Hm, Hld, Hdi);
  for (int i = 0; i < nv; i++) qacc[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i];
  solve_ld(m, Hld, Hdi, qacc);
  free(Hld);
  for (int u = 0; u < m->na; u++) d->act[u] += dt * d->act_dot[u];
  for (int i = 0; i < nv; i++) d->qvel[i] += dt * qacc[i];
  for (int j = 0; j < m->njnt; j++) {
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    if (m->jnt_type[j] == JNT_FREE) {
      for (int k = 0; k < 3; k++) d->qpos[qa + k] += dt * d->qvel[da + k];
      real w[3] = {d->qvel[da + 3], d->qvel[da + 4], d->qvel[da + 5]};
      real n = SQRT(dot3(w, w));
      real ax[3] = {0, 0, 0};
      if (n > MINVAL) for (int k = 0; k < 3; k++) ax[k] = w[k] / n;
      real qr[4], qn[4];
      axis_angle_quat(qr, ax, dt * n);
      quat_mul(qn, d->qpos + qa + 3, qr);
      quat_normalize(qn);
      memcpy(d->qpos + qa + 3, qn, sizeof(qn));
    } else {
      d->qpos[qa] += dt * d->qvel[da];
    }
  }
  d->time += dt;
}

/* pipeline_step [UP brax.mjx.pipeline.step x n_frames]: ctrl is held over the substeps */
void ref_step(const ref_model* m, ref_data* d, const double* ctrl, int n_frames) {
  for (int u = 0; u < m->nu; u++) d->ctrl[u] = (real)ctrl[u];
  for (int f = 0; f < n_frames; f++) {
    ref_forward(m, d);
    euler(m, d);
  }
}

/* pipeline_init [UP brax.mjx.pipeline.init]: make_data + set qpos/qvel + forward */
void ref_init(const ref_model* m, ref_data* d, const double* qpos, const double* qvel) {
  for (int i = 0; i < m->nq; i++) d->qpos[i] = (real)qpos[i];
  for (int i = 0; i < m->nv; i++) { d->qvel[i] = (real)qvel[i]; d->qacc_warmstart[i] = 0; }
  for (int u = 0; u < m->na; u++) d->act[u] = 0;
  for (int u = 0; u < m->nu; u++) d->ctrl[u] = 0;
  d->time = 0;
  ref_forward(m, d);
}

/* ------------------------------------------------------------------ field access (tests) */
typedef struct { const char* name; real* p; size_t n; } field;
static int fields(const ref_model* m, ref_data* d, field* f) {
  int nb = m->nbody, nv = m->nv, k = 0;
#define F(nm, cnt) f[k].name = #nm, f[k].p = d->nm, f[k].n = (cnt), k++
  F(qpos, m->nq); F(qvel, nv); F(act, m->na); F(ctrl, m->nu); F(qacc_warmstart, nv);
  F(xpos, nb * 3); F(xquat, nb * 4); F(xmat, nb * 9); F(xipos, nb * 3); F(ximat, nb * 9);
  F(xanchor, m->njnt * 3); F(xaxis, m->njnt * 3); F(subtree_com, nb * 3); F(geom_xpos, m->ngeom * 3);
  F(geom_xmat, m->ngeom * 9); F(cinert, nb * 10); F(crb, nb * 10); F(cdof, nv * 6); F(cdof_dot, nv * 6);
  F(cvel, nb * 6); F(cacc, nb * 6); F(cfrc, nb * 6); F(qM, m->nM); F(qLD, m->nM); F(qLDiagInv, nv);
  F(con_dist, m->ncon); F(con_pos, m->ncon * 3); F(con_frame, m->ncon * 9);
  F(efc_J, (size_t)m->nefc * nv); F(efc_D, m->nefc); F(efc_aref, m->nefc); F(efc_force, m->nefc); F(efc_pos, m->nefc);
  F(qfrc_passive, nv); F(qfrc_bias, nv); F(qfrc_actuator, nv); F(qfrc_smooth, nv); F(qacc_smooth, nv);
  F(qfrc_constraint, nv); F(qacc, nv); F(act_dot, m->na); F(actuator_force, m->nu);
  return k;
}
long ref_get(const ref_model* m, ref_data* d, const char* name, double* out, long cap) {
  field f[64];
  int n = fields(m, d, f);
  if (!strcmp(name, "solver_niter")) { if (cap > 0) out[0] = d->solver_niter; return 1; }
  if (!strcmp(name, "solver_cost")) { if (cap > 0) out[0] = (double)d->solver_cost; return 1; }
  if (!strcmp(name, "time")) { if (cap > 0) out[0] = (double)d->time; return 1; }
  for (int i = 0; i < n; i++)
    if (!strcmp(f[i].name, name)) {
      for (size_t k = 0; k < f[i].n && (long)k < cap; k++) out[k] = (double)f[i].p[k];
      return (long)f[i].n;
    }
  return -1;
}
long ref_set(const ref_model* m, ref_data* d, const char* name, const double* in, long cnt) {
  field f[64];
  int n = fields(m, d, f);
  for (int i = 0; i < n; i++)
    if (!strcmp(f[i].name, name)) {
      for (size_t k = 0; k < f[i].n && (long)k < cnt; k++) f[i].p[k] = (real)in[k];
      return (long)f[i].n;
    }
  return -1;
}

/* ------------------------------------------------------------------ env layer [REF Rodent_Env_Brax.py:98-162] */
/* obs = [qpos, qvel, cinert[1:], cvel[1:], qfrc_actuator, xmat[1] @ (track_pos[frame+1] - qpos[:3])] */
void ref_get_obs(const ref_model* m, const ref_data* d, const double* track_pos, int T, int cur_frame, double* obs) {
  int k = 0, nb = m->nbody;
  for (int i = 0; i < m->nq; i++) obs[k++] = (double)d->qpos[i];
  for (int i = 0; i < m->nv; i++) obs[k++] = (double)d->qvel[i];
  for (int i = 10; i < 10 * nb; i++) obs[k++] = (double)d->cinert[i];
  for (int i = 6; i < 6 * nb; i++) obs[k++] = (double)d->cvel[i];
  for (int i = 0; i < m->nv; i++) obs[k++] = (double)d->qfrc_actuator[i];
  int fi = cur_frame + 1;
  if (fi < 0) fi = 0;
  if (fi > T - 1) fi = T - 1; /* JAX clamps out-of-range gather indices */
  real v[3], r[3];
  for (int i = 0; i < 3; i++) v[i] = (real)track_pos[3 * fi + i] - d->qpos[i];
  mat_vec(r, d->xmat + 9, v);
  for (int i = 0; i < 3; i++) obs[k++] = (double)r[i];
}

/* one env step: returns reward, done; updates cur_frame; metrics = {pos_reward, reward_quadctrl, reward_alive} */
void ref_env_step(const ref_model* m, ref_data* d, const double* action, int n_frames, const double* track_pos, int T,
                  int* cur_frame, double healthy_reward, double ctrl_cost_weight, double min_z, double max_z,
                  int terminate_when_unhealthy, double* obs, double* reward, double* done, double* metrics) {
  ref_step(m, d, action, n_frames);
  int old = *cur_frame;
  int fi = old < 0 ? 0 : (old > T - 1 ? T - 1 : old);
  real dx[3];
  for (int i = 0; i < 3; i++) dx[i] = d->qpos[i] - (real)track_pos[3 * fi + i];
  real pos_reward = EXP(R(-100.0) * SQRT(dot3(dx, dx)));
  real z = d->qpos[2];
  real healthy = z < (real)min_z ? 0 : 1;
  if (z > (real)max_z) healthy = 0;
  real hr = terminate_when_unhealthy ? (real)healthy_reward : (real)healthy_reward * healthy;
  real cc = 0;
  for (int u = 0; u < m->nu; u++) cc += (real)action[u] * (real)action[u];
  cc *= (real)ctrl_cost_weight;
  *cur_frame = old + 1;
  ref_get_obs(m, d, track_pos, T, *cur_frame, obs);
  *reward = (double)(pos_reward + hr - cc);
  *done = terminate_when_unhealthy ? (double)(1 - healthy) : 0.0;
  metrics[0] = (double)pos_reward; metrics[1] = (double)-cc; metrics[2] = (double)hr;
}

/* ------------------------------------------------------------------ batched helpers (cpu_baseline timing, long parity runs) */
void ref_env_step_batch(const ref_model* m, ref_data** ds, const double* action /*[N][nu]*/, int N, int n_frames, const double* track_pos, int T,
                        int* cur_frame /*[N]*/, double healthy_reward, double ctrl_cost_weight, double min_z, double max_z,
                        int terminate_when_unhealthy, double* obs /*[N][obs_dim]*/, double* reward, double* done, double* metrics /*[N][3]*/) {
#pragma omp parallel for schedule(dynamic, 1)
  for (int e = 0; e < N; e++)
    ref_env_step(m, ds[e], action + (size_t)e * m->nu, n_frames, track_pos, T, cur_frame + e, healthy_reward, ctrl_cost_weight, min_z, max_z,
                 terminate_when_unhealthy, obs + (size_t)e * m->obs_dim, reward + e, done + e, metrics + 3 * (size_t)e);
}
void ref_step_batch(const ref_model* m, ref_data** ds, const double* ctrl /*[N][nu]*/, int N, int n_frames) {
#pragma omp parallel for schedule(dynamic, 4)
  for (int e = 0; e < N; e++) ref_step(m, ds[e], ctrl + (size_t)e * m->nu, n_frames);
}
