// rr_api.hip -- host side of librodent_hip.so: model blob loader, device upload of the kernel
// tables, LDS / debug layouts and the launch wrappers behind the C ABI of include/rodent_rr.h.
#include "../../include/rodent_rr.h"
#include "rr_kernel.h"
#include "rr_mlp.h"
#include "rr_ppo.h"

#include <atomic>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
extern "C" const char* rr_last_error(void) { return g_err.c_str(); }

#define HIPCHK(x)                                                                                  \
  do {                                                                                             \
    hipError_t e_ = (x);                                                                           \
    if (e_ != hipSuccess) return fail(RR_EHIP, std::string(#x) + ": " + hipGetErrorString(e_));   \
  } while (0)

typedef void (*kern_t)(const RRDims, const RRTables, const RRIO, const int, const int);
// ------------------------------------------------------------------------------------------ model
struct Entry { int dtype, ndim, dims[4]; size_t count; const void* data; };

struct rr_model {
  std::vector<unsigned char> raw;
  std::map<std::string, Entry> e;
  rr_dims dims;
  RRDims kd;
  std::vector<std::string> dbg_names;
  std::vector<int32_t> dbg_off, dbg_size;
  std::vector<const char*> dbg_cnames;
  int NBS, NVS, NCS;
  bool stage_ok = true;
  int solver = 1;          // 1 = CG, 2 = Newton [REF Rodent_Env_Brax.py:42-45]
  // two identical trees (rodent_pair.xml): dims / LDS layout of ONE replica for the two-wave instance (rr_kernel.h PAIR), from the
  // blob's h_* tables; pair_ok = the blob carries them and a PAIR kernel instance matches
  bool pair_ok = false;
  RRDims kd_rep;
  bool dyn = false;        // candidate-pair contacts between moving bodies / condim 1 / tendon transmissions (rr_kernel.h DYN; blob k_dyn)
  mutable std::atomic<int> live_batches{0};     // batches bake the model's LDS layout at creation: the solver type is fixed while any exist

  const Entry* find(const char* n) const { auto it = e.find(n); return it == e.end() ? nullptr : &it->second; }
  int iscalar(const char* n) const { const Entry* x = find(n); return x ? ((const int32_t*)x->data)[0] : 0; }
  float fscalar(const char* n, int i = 0) const { const Entry* x = find(n); return x ? ((const float*)x->data)[i] : 0.f; }
};

static void layout(rr_model* m) {
  RRDims& k = m->kd;
  const rr_dims& d = m->dims;
  const int con_slots = m->dyn ? m->NCS * RR_LANES : d.ncon;      // DYN: the wave holds the pairs in penetration in its contact slots
  const RRLayout L = rr_layout(d.nq, d.nv, d.nu, d.nbody, d.nM, con_slots, m->solver == 2);
  k.o_H = L.o_H; k.o_Mp = L.o_Mp; k.o_anc = L.o_anc; k.solver = m->solver;
  // factorisation schedule: with alias copies of the hot rows in the pose cells (levelsched.py); the Newton instances reuse the schedule
  // on the Hessian's array through an address shift, which the alias cells would not follow, so they run the alias-free one
  k.nfac = m->iscalar(m->solver == 2 ? "k_factor3p_rows" : "k_factor3_rows");
  k.nalias = m->solver == 2 ? 0 : m->iscalar("k_nalias");
  m->dbg_names.clear(); m->dbg_off.clear(); m->dbg_size.clear(); m->dbg_cnames.clear();      // layout() may run again (rr_model_set_solver_type)
  k.o_qpos = L.o_qpos; k.o_qvel = L.o_qvel; k.o_act = L.o_act; k.o_ctrl = L.o_ctrl; k.o_xpos = L.o_xpos; k.o_xquat = L.o_xquat;
  k.o_cinert = L.o_cinert; k.o_cdof = L.o_cdof; k.o_cvel = L.o_cvel; k.o_qLD = L.o_qLD; k.o_vec = L.o_vec; k.o_x = L.o_x;
  k.o_arm = L.o_arm; k.o_warm = L.o_warm; k.o_qact = L.o_qact; k.o_jlist = L.o_jlist; k.lds_floats = L.lds_floats;
  const int o = L.lds_floats;
  // staging of the line search's compacted rows: cinert | cvel | pose regions each hold 4*ncon + nv floats
  m->stage_ok = 2 * (k.nalias + 4) <= std::max(7 * d.nbody + 4, 6 * d.nv) && 2 * (d.nbody + 8) <= std::max(7 * d.nbody + 4, 6 * d.nv) && 6 * con_slots <= std::max(7 * d.nbody + 4, 6 * d.nv) && 4 * con_slots + d.nv <= std::min(std::min(10 * d.nbody, 6 * d.nbody), std::max(7 * d.nbody + 4, 6 * d.nv));
  if (m->dyn && d.nu > std::max(7 * d.nbody + 4, 6 * d.nv)) m->stage_ok = false;      // actuator forces go through the pose cells
  // debug dump
  int g = 0;
  auto dbg = [&](const char* name, int n) { int r = g; m->dbg_names.push_back(name); m->dbg_off.push_back(g); m->dbg_size.push_back(n); g += n; return r; };
  k.g_xpos = dbg("xpos", 3 * d.nbody); k.g_xquat = dbg("xquat", 4 * d.nbody); k.g_xmat = dbg("xmat", 9 * d.nbody);
  k.g_com = dbg("subtree_com", 6); k.g_cinert = dbg("cinert", 10 * d.nbody); k.g_crb = dbg("crb", 10 * d.nbody);
  k.g_cdof = dbg("cdof", 6 * d.nv); k.g_cvel = dbg("cvel", 6 * d.nbody); k.g_cfrc = dbg("cfrc", 6 * d.nbody);
  k.g_qM = dbg("qM", d.nM); k.g_qLD = dbg("qLD", d.nM); k.g_dinv = dbg("qLDiagInv", d.nv);
  k.g_bias = dbg("qfrc_bias", d.nv); k.g_passive = dbg("qfrc_passive", d.nv); k.g_actuator = dbg("qfrc_actuator", d.nv);
  k.g_smooth = dbg("qfrc_smooth", d.nv); k.g_qacc_smooth = dbg("qacc_smooth", d.nv);
  k.g_con_dist = dbg("con_dist", d.ncon); k.g_con_pos = dbg("con_pos", 3 * d.ncon); k.g_con_frame = dbg("con_frame", 9 * d.ncon);
  k.g_con_D = dbg("con_D", d.ncon); k.g_con_aref = dbg("con_aref", 4 * d.ncon); k.g_lim = dbg("limit_pos_D_aref", 3 * d.nv);
  k.g_qacc = dbg("qacc", d.nv); k.g_qfrc_constraint = dbg("qfrc_constraint", d.nv); k.g_misc = dbg("niter_cost", 2);
  k.g_kaok = dbg("kernarg_ok", 1);
  k.dbg_floats = g;
  for (auto& s : m->dbg_names) m->dbg_cnames.push_back(s.c_str());
  m->dims.lds_bytes = o * (int)sizeof(float);
  m->dims.dbg_floats = g;
  m->dims.solver = m->solver;
  // 1: the production launches of this model run an instance compiled for its dimensions (rr_kernel.h RRDimsFixed); 0: the generic
  // instance (same results, ~10 % slower).  The constants cover the schedule-table parameters, so a compiler change that moves
  // them shows up here (tests/test_abi_and_oracle.py) instead of as a silent slowdown.
  m->dims.fixed_instance = (m->solver != 2 && !m->dyn && m->NBS == 2 && m->NVS == 2 && m->NCS == 1 && (RRDimsRodent::matches(k) || RRDimsRodentNew::matches(k))) ? 1 : 0;
}

static kern_t pick_pair_kernel(const rr_model* m);
// Two-wave (PAIR) instance: dims of one replica from the blob's h_* tables (rodent_amd/ktables.py replica_model).  RR_PAIR_WAVES=0
// keeps such models on the generic one-wave instance (A/B runs).
static void setup_replica(rr_model* m) {
  m->pair_ok = false;
  const char* sw = getenv("RR_PAIR_WAVES");
  if (sw && sw[0] == '0') return;
  static const char* need[] = {"h_dims", "h_k_slots", "h_k_body_i", "h_k_body_f", "h_k_jnt_i", "h_k_jnt_f", "h_k_dof_i", "h_k_dof_f", "h_k_act_f", "h_k_M_ij_k",
                               "h_k_body_anc", "h_k_nround", "h_k_factor3", "h_k_factor3_rows", "h_k_linv", "h_k_linv_rows", "h_k_coljob", "h_k_rowjob", "h_k_jobown",
                               "h_k_solve_lmax", "h_k_solve_cmax", "h_k_solve_rmax", "h_k_nalias", "h_k_con_i", "h_k_con_f", "h_k_con_chain_rows", "h_k_root_mass"};
  for (const char* n : need) if (!m->find(n)) return;
  const int32_t* hd = (const int32_t*)m->find("h_dims")->data;       // nq nv nu nbody njnt nM ncon dmax
  const int32_t* sl = (const int32_t*)m->find("h_k_slots")->data;
  if (m->find("h_dims")->count < 8 || sl[0] != 2 || sl[1] != 2 || sl[2] != 1) return;
  if (2 * hd[0] != m->dims.nq || 2 * hd[1] != m->dims.nv || 2 * hd[2] != m->dims.nu || 2 * hd[6] != m->dims.ncon || m->dims.na != m->dims.nu) return;
  RRDims k = m->kd;          // options, gravity, tolerances, meaninertia: the model's
  k.nq = hd[0]; k.nv = hd[1]; k.nu = hd[2]; k.nbody = hd[3]; k.njnt = hd[4]; k.nM = hd[5]; k.ncon = hd[6]; k.dmax = hd[7];
  k.nroot = (int)m->find("h_k_root_mass")->count;
  k.nround = m->iscalar("h_k_nround"); k.ninv = m->iscalar("h_k_linv_rows"); k.nfac = m->iscalar("h_k_factor3_rows"); k.nalias = m->iscalar("h_k_nalias");
  k.lmax = m->iscalar("h_k_solve_lmax"); k.cmax = m->iscalar("h_k_solve_cmax"); k.rmax = m->iscalar("h_k_solve_rmax");
  k.obs_dim = k.nq + k.nv + 16 * (k.nbody - 1) + k.nv + 3;
  const RRLayout L = rr_layout(k.nq, k.nv, k.nu, k.nbody, k.nM, k.ncon, false);
  k.o_qpos = L.o_qpos; k.o_qvel = L.o_qvel; k.o_act = L.o_act; k.o_ctrl = L.o_ctrl; k.o_xpos = L.o_xpos; k.o_xquat = L.o_xquat;
  k.o_cinert = L.o_cinert; k.o_cdof = L.o_cdof; k.o_cvel = L.o_cvel; k.o_qLD = L.o_qLD; k.o_vec = L.o_vec; k.o_x = L.o_x;
  k.o_arm = L.o_arm; k.o_warm = L.o_warm; k.o_qact = L.o_qact; k.o_jlist = L.o_jlist; k.lds_floats = L.lds_floats;
  k.o_H = k.o_Mp = k.o_anc = 0; k.solver = 1;
  k.nv_scale = m->dims.nv;                          // the solver's tolerances scale with the MODEL's dof count
  k.lds_bytes_rep = L.lds_floats * 4;
  k.fac_stride = (int)m->find("h_k_factor3")->count; k.inv_stride = (int)m->find("h_k_linv")->count;
  const int njs = 2 * RR_LANES;
  if ((int)m->find("h_k_coljob")->count != 9 * njs || (int)m->find("h_k_rowjob")->count != 5 * njs || k.lmax > 16 || k.lmax < 1 || k.cmax > 8 || k.rmax > 8 || k.cmax < 1) return;
  if (k.nM > RR_LANES * 18 || k.nroot != 1) return;
  const bool stage_ok = 2 * (k.nbody + 8) <= std::max(7 * k.nbody + 4, 6 * k.nv) && 6 * k.ncon <= std::max(7 * k.nbody + 4, 6 * k.nv) &&
                        4 * k.ncon + k.nv <= std::min(std::min(10 * k.nbody, 6 * k.nbody), std::max(7 * k.nbody + 4, 6 * k.nv));
  if (!stage_ok || 2 * (k.nalias + 4) > std::max(7 * k.nbody + 4, 6 * k.nv) || 2 * k.lds_bytes_rep + 128 > 64 * 1024) return;
  m->kd_rep = k;
  m->pair_ok = pick_pair_kernel(m) != nullptr;
}

extern "C" int rr_model_load(const char* path, rr_model** out) {
  if (!path || !out) return fail(RR_EINVAL, "rr_model_load: null argument");
  FILE* f = fopen(path, "rb");
  if (!f) return fail(RR_EIO, std::string("rr_model_load: cannot open ") + path);
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  rr_model* m = new rr_model();
  m->raw.resize(n);
  size_t got = fread(m->raw.data(), 1, n, f);
  fclose(f);
  if ((long)got != n || n < 8 || memcmp(m->raw.data(), "RRM1", 4)) { delete m; return fail(RR_EIO, "rr_model_load: not an RRM1 blob"); }
  const unsigned char* raw = m->raw.data();
  uint32_t ne = *(const uint32_t*)(raw + 4);
  if (8 + (size_t)ne * 72 > (size_t)n) { delete m; return fail(RR_EIO, "rr_model_load: truncated header"); }
  const unsigned char* p = raw + 8;
  for (uint32_t i = 0; i < ne; ++i, p += 72) {
    Entry e;
    char name[33] = {0};
    memcpy(name, p, 32);
    e.dtype = *(const uint32_t*)(p + 32);
    e.ndim = *(const uint32_t*)(p + 36);
    for (int k = 0; k < 4; ++k) e.dims[k] = *(const uint32_t*)(p + 40 + 4 * k);
    uint64_t off = *(const uint64_t*)(p + 56), nb = *(const uint64_t*)(p + 64);
    if (off + nb > (uint64_t)n) { delete m; return fail(RR_EIO, "rr_model_load: entry out of range"); }
    e.count = nb / 4;
    e.data = raw + off;
    m->e[name] = e;
  }
  static const char* need[] = {"nq", "nv", "nu", "nbody", "njnt", "ngeom", "nM", "ncon", "nlimit", "nefc", "obs_dim", "k_slots",
                               "k_body_i", "k_body_f", "k_jnt_i", "k_jnt_f", "k_dof_i", "k_dof_f", "k_act_f", "k_M_ij_k", "k_body_anc",
                               "k_nround", "k_factor3", "k_factor3_rows", "k_factor3p", "k_factor3p_rows", "k_nalias", "k_linv", "k_linv_rows", "k_coljob", "k_rowjob",
                               "k_jobown", "k_solve_lmax", "k_solve_cmax", "k_solve_rmax", "k_con_i", "k_con_f", "k_con_chain_packed", "k_con_chain_rows", "k_root_mass",
                               "k_act_i", "k_act_m_i", "k_act_m_f",
                               "dof_depth", "body_depth", "opt_timestep", "opt_gravity", "stat_meaninertia"};
  for (const char* nme : need)
    if (!m->find(nme)) { std::string s = std::string("rr_model_load: blob lacks '") + nme + "'"; delete m; return fail(RR_EIO, s); }
  rr_dims& d = m->dims;
  memset(&d, 0, sizeof(d));
  d.nq = m->iscalar("nq"); d.nv = m->iscalar("nv"); d.nu = m->iscalar("nu"); d.na = m->iscalar("na"); d.nbody = m->iscalar("nbody");
  d.njnt = m->iscalar("njnt"); d.ngeom = m->iscalar("ngeom"); d.nM = m->iscalar("nM"); d.ncon = m->iscalar("ncon");
  d.nlimit = m->iscalar("nlimit"); d.nefc = m->iscalar("nefc"); d.obs_dim = m->iscalar("obs_dim");
  d.iterations = m->iscalar("opt_iterations"); d.ls_iterations = m->iscalar("opt_ls_iterations");
  d.timestep = m->fscalar("opt_timestep");
  const int32_t* sl = (const int32_t*)m->find("k_slots")->data;
  m->NBS = sl[0]; m->NVS = sl[1]; m->NCS = sl[2];
  m->dyn = m->iscalar("k_dyn") != 0;
  if (m->dyn && (m->find("k_con_i")->dims[1] != 8 || m->find("k_con_f")->dims[1] != 32 || (int)m->find("k_con_chain_rows")->count != 10 * (d.ncon + 1) || d.nv >= 128 || d.nu > RR_LANES)) {
    delete m; return fail(RR_EIO, "rr_model_load: candidate-pair tables do not match the DYN kernel instance (stale blob)"); }
  RRDims& k = m->kd;
  memset(&k, 0, sizeof(k));
  k.nq = d.nq; k.nv = d.nv; k.nu = d.nu; k.nbody = d.nbody; k.njnt = d.njnt; k.nM = d.nM; k.ncon = d.ncon;
  int dmax = 0;
  { const Entry* dd = m->find("dof_depth"); for (size_t i = 0; i < dd->count; ++i) dmax = std::max(dmax, ((const int32_t*)dd->data)[i]); }
  k.dmax = dmax;
  k.nroot = (int)m->find("k_root_mass")->count;
  k.nround = m->iscalar("k_nround");
  k.ninv = m->iscalar("k_linv_rows");
  if (d.nM > RR_LANES * (m->NVS == 1 ? 10 : (m->NVS == 2 ? 18 : 35))) { delete m; return fail(RR_EUNSUPPORTED, "rr_model_load: more mass-matrix entries than the kernel's register table"); }
  { const int njs = (m->NVS >= 3 ? m->NVS + 1 : m->NVS) * RR_LANES;     // Wave::NJS job slots
    k.lmax = m->iscalar("k_solve_lmax"); k.cmax = m->iscalar("k_solve_cmax"); k.rmax = m->iscalar("k_solve_rmax");
    if ((int)m->find("k_coljob")->count != 9 * njs || (int)m->find("k_rowjob")->count != 5 * njs || k.lmax > 16 || k.lmax < 1 || k.cmax > 8 || k.rmax > 8 || k.cmax < 1) {
      delete m; return fail(RR_EIO, "rr_model_load: solve job tables do not match the kernel instance (stale blob)"); } }
  k.obs_dim = d.obs_dim; k.iterations = d.iterations; k.ls_iterations = d.ls_iterations;
  k.dt = d.timestep; k.gx = m->fscalar("opt_gravity", 0); k.gy = m->fscalar("opt_gravity", 1); k.gz = m->fscalar("opt_gravity", 2);
  k.tolerance = m->fscalar("opt_tolerance"); k.ls_tolerance = m->fscalar("opt_ls_tolerance");
  k.meaninertia = m->fscalar("stat_meaninertia");
  if (k.nroot > 2) { delete m; return fail(RR_EUNSUPPORTED, "rr_model_load: more than 2 kinematic trees"); }
  if (m->find("k_dof_i")->dims[1] != RR_DOFI) { delete m; return fail(RR_EIO, "rr_model_load: k_dof_i width mismatch (stale blob)"); }
  if (m->find("k_body_i")->dims[1] != RR_BODYI) { delete m; return fail(RR_EIO, "rr_model_load: k_body_i width mismatch (stale blob)"); }
  if (d.nv > 256 || (!m->dyn && d.ncon > 256)) { delete m; return fail(RR_EUNSUPPORTED, "rr_model_load: nv/ncon above the 8-bit table index"); }
  if (m->find("k_con_chain_packed")->dims[1] != m->NCS * RR_LANES) { delete m; return fail(RR_EIO, "rr_model_load: lane-table width mismatch"); }
  k.nv_scale = d.nv > 1 ? d.nv : 1;
  layout(m);
  setup_replica(m);
  *out = m;
  return RR_OK;
}

extern "C" int rr_model_dims(const rr_model* m, rr_dims* out) {
  if (!m || !out) return fail(RR_EINVAL, "rr_model_dims: null argument");
  *out = m->dims;
  return RR_OK;
}
extern "C" int rr_model_set_solver(rr_model* m, int32_t it, int32_t ls) {
  if (!m || it < 0 || ls < 0) return fail(RR_EINVAL, "rr_model_set_solver: bad argument");
  m->dims.iterations = m->kd.iterations = it;
  m->dims.ls_iterations = m->kd.ls_iterations = ls;
  return RR_OK;
}
extern "C" int rr_model_set_solver_type(rr_model* m, int32_t solver) {
  if (!m || (solver != 1 && solver != 2)) return fail(RR_EINVAL, "rr_model_set_solver_type: solver must be 1 (cg) or 2 (newton)");
  if (solver != m->solver && m->live_batches.load() > 0)
    return fail(RR_EINVAL, "rr_model_set_solver_type: batches of this model exist (they hold the LDS layout of the current solver); set the solver before rr_batch_create");
  if (solver == 2 && (!(m->NBS == 2 && m->NVS == 2 && m->NCS == 1) || m->dyn))
    return fail(RR_EUNSUPPORTED, "rr_model_set_solver_type: the Newton instance exists for the single-rodent models only");
  if (solver == 2 && 4 * ((m->dims.nv + 3) & ~3) > std::max(7 * m->dims.nbody + 4, 6 * m->dims.nv))
    return fail(RR_EUNSUPPORTED, "rr_model_set_solver_type: the four Hessian rows do not fit the pose cells");
  m->solver = solver;
  layout(m);
  return RR_OK;
}
extern "C" void rr_model_destroy(rr_model* m) { delete m; }
extern "C" int rr_model_table(const rr_model* m, const char* name, const void** host_ptr, size_t* count, int32_t* dtype) {
  if (!m || !name || !host_ptr) return fail(RR_EINVAL, "rr_model_table: null argument");
  const Entry* e = m->find(name);
  if (!e) return fail(RR_EINVAL, std::string("rr_model_table: the model has no table '") + name + "'");
  *host_ptr = e->data;
  if (count) *count = e->count;
  if (dtype) *dtype = e->dtype;
  return RR_OK;
}
extern "C" int rr_kernarg_layout(int32_t* io_offset, int32_t* io_size, int32_t* total_size) {
  if (io_offset) *io_offset = (int32_t)offsetof(RRKArgs, io);
  if (io_size) *io_size = (int32_t)sizeof(RRIO);
  if (total_size) *total_size = (int32_t)sizeof(RRKArgs);
  return RR_OK;
}

// ------------------------------------------------------------------------------------------ batch
struct rr_batch {
  const rr_model* m;
  int N, device;
  hipStream_t stream;
  RRDims kd;
  RRTables T;
  RRTables T_rep;                       // tables of one replica (two-wave instance of a two-tree model; m->pair_ok)
  std::vector<void*> dev_allocs;
  bool timing = false;
  std::vector<hipEvent_t> ev0, ev1;     // ring of event pairs: launches are timed without a host sync per launch
  int npending = 0, ring_head = 0;      // pending pairs are ring_head .. ring_head + npending - 1 (mod the ring size)
  double total_ms = 0;
  int64_t launches = 0;
  unsigned long long* prof = nullptr;   // diagnostic phase-cycle buffer (rr_batch_set_profile)
  const int32_t* env_map = nullptr;     // rr_batch_set_schedule
  uint32_t* cost = nullptr;
  bool counted = false;                 // this batch is in m->live_batches
  unsigned* dyn_overflow = nullptr;     // DYN models: launches x envs in which more pairs penetrated than the wave has contact slots (rr_batch_contact_overflow)
  unsigned* progress = nullptr;         // pacing counter of multi-step launches (RRIO::progress); RR_PACE=0 turns pacing off
};

template <typename Ptr>
static int upload(rr_batch* b, const char* name, Ptr* dst) {
  const Entry* e = b->m->find(name);
  void* p = nullptr;
  size_t bytes = std::max<size_t>(e->count, 1) * 4;
  HIPCHK(hipMalloc(&p, bytes));
  b->dev_allocs.push_back(p);
  if (e->count) HIPCHK(hipMemcpy(p, e->data, e->count * 4, hipMemcpyHostToDevice));
  *dst = (Ptr)p;
  return RR_OK;
}

// Row schedules (k_factor3, k_linv): element indices -> LDS byte addresses (rr_kernel.h run_levels).  Indices below nM + 4 are cells of
// the sparse-matrix array at `base`; the alias cells nM + 4 .. live at `alias_base` (the pose cells).
// `copies` = 2 (two-wave instance): a second copy behind the first, addressed into the second wavefront's LDS region (+ rep_bytes).
static int upload_levels(rr_batch* b, const char* name, rr_gi* dst, uint32_t base, uint32_t alias_base, uint32_t nM, int copies = 1, uint32_t rep_bytes = 0) {
  const Entry* e = b->m->find(name);
  std::vector<uint32_t> t((size_t)copies * e->count);
  for (int c = 0; c < copies; ++c) {
    const uint32_t* src = (const uint32_t*)e->data;
    uint32_t* tc = t.data() + (size_t)c * e->count;
    const uint32_t cb = base + (uint32_t)c * rep_bytes, ab = alias_base + (uint32_t)c * rep_bytes;
    for (size_t i = 0; i + 3 < e->count; i += 4) {        // x = a | b0 << 16, y = d0 | d1 << 16, z = d2 | d3 << 16, w = q
      uint32_t f[6] = {src[i] & 0xFFFFu, src[i] >> 16, src[i + 1] & 0xFFFFu, src[i + 1] >> 16, src[i + 2] & 0xFFFFu, src[i + 2] >> 16};
      for (uint32_t& v : f) {                              // 8 bytes per entry: pairs
        v = v < nM + 4u ? v * 8u + cb : (v - (nM + 4u)) * 8u + ab;
        if (v > 0xFFFFu) return fail(RR_EUNSUPPORTED, "level schedule does not fit the 16-bit LDS address fields");
      }
      const uint32_t q = src[i + 3];
      if (q > 255u) return fail(RR_EIO, "level schedule: pivot offset out of range (stale blob)");
      tc[i] = f[0] | (f[1] << 16); tc[i + 1] = f[2] | (f[3] << 16); tc[i + 2] = f[4] | (f[5] << 16);
      tc[i + 3] = q * 8u;
    }
  }
  void* p = nullptr;
  HIPCHK(hipMalloc(&p, t.size() * 4));
  b->dev_allocs.push_back(p);
  HIPCHK(hipMemcpy(p, t.data(), t.size() * 4, hipMemcpyHostToDevice));
  *dst = (rr_gi)p;
  return RR_OK;
}

// Instances: production (no debug dump; fixed-dimension variant for the rodent dims), debug dump (generic dims; selected per
// launch when rr_outputs.debug is given), cycle-stamp profile (diagnostic, rodent dims or generic 2,2,1).
static kern_t pick_kernel(const rr_model* m, bool prof = false, bool dbg = false) {
  const int nbs = m->NBS, nvs = m->NVS, ncs = m->NCS;
  if (m->dyn)        // candidate-pair contacts: one production instance (generic dimensions); no debug dump, no profile build
    return (!prof && !dbg && m->solver != 2 && nbs == 2 && nvs == 2 && ncs == 1) ? rr_step_kernel<2, 2, 1, false, false, RRDims, false, false, false, false, true> : nullptr;
  if (prof && m->solver != 2) return (nbs == 2 && nvs == 2 && ncs == 1) ? (RRDimsRodent::matches(m->kd) ? rr_step_kernel<2, 2, 1, true, false, RRDimsRodent> : rr_step_kernel<2, 2, 1, true, false, RRDims>) : nullptr;
  if (m->solver == 2)     // Newton: the (2,2,1) generic instances only (rr_model_set_solver_type checks)
    return prof ? nullptr : (dbg ? rr_step_kernel<2, 2, 1, false, true, RRDims, true> : rr_step_kernel<2, 2, 1, false, false, RRDims, true>);
  if (dbg) {
    if (nbs == 1 && nvs == 1 && ncs == 1) return rr_step_kernel<1, 1, 1, false, true, RRDims>;
    if (nbs == 2 && nvs == 2 && ncs == 1) return rr_step_kernel<2, 2, 1, false, true, RRDims>;
    if (nbs == 3 && nvs == 3 && ncs == 2) return rr_step_kernel<3, 3, 2, false, true, RRDims>;
    return nullptr;
  }
  if (nbs == 2 && nvs == 2 && ncs == 1 && RRDimsRodent::matches(m->kd)) return rr_step_kernel<2, 2, 1, false, false, RRDimsRodent>;   // fixed-dimension instances
  if (nbs == 2 && nvs == 2 && ncs == 1 && RRDimsRodentNew::matches(m->kd)) return rr_step_kernel<2, 2, 1, false, false, RRDimsRodentNew>;
  if (nbs == 1 && nvs == 1 && ncs == 1) return rr_step_kernel<1, 1, 1, false, false, RRDims>;
  if (nbs == 2 && nvs == 2 && ncs == 1) return rr_step_kernel<2, 2, 1, false, false, RRDims>;
  if (nbs == 3 && nvs == 3 && ncs == 2) return rr_step_kernel<3, 3, 2, false, false, RRDims>;
  return nullptr;
}

static kern_t pick_unroll_kernel(const rr_model* m, bool actor = false) {
  if (m->solver == 2 || m->dyn || !(m->NBS == 2 && m->NVS == 2 && m->NCS == 1)) return nullptr;
  if (actor) return RRDimsRodent::matches(m->kd) ? rr_step_kernel<2, 2, 1, false, false, RRDimsRodent, false, true, true>
                    : (RRDimsRodentNew::matches(m->kd) ? rr_step_kernel<2, 2, 1, false, false, RRDimsRodentNew, false, true, true> : rr_step_kernel<2, 2, 1, false, false, RRDims, false, true, true>);
  return RRDimsRodent::matches(m->kd) ? rr_step_kernel<2, 2, 1, false, false, RRDimsRodent, false, true>
         : (RRDimsRodentNew::matches(m->kd) ? rr_step_kernel<2, 2, 1, false, false, RRDimsRodentNew, false, true> : rr_step_kernel<2, 2, 1, false, false, RRDims, false, true>);
}
// the two-wave instance of a two-tree model (one replica = the rodent_new dims); nullptr: no such instance -> generic one-wave kernel
static kern_t pick_pair_kernel(const rr_model* m) {
  if (m->solver == 2) return nullptr;
  return RRDimsRodentNew::matches(m->kd_rep) ? rr_step_kernel<2, 2, 1, false, false, RRDimsRodentNew, false, false, false, true> : nullptr;
}

extern "C" int rr_batch_create(const rr_model* m, int32_t num_envs, int32_t device, void* stream, rr_batch** out) {
  if (!m || !out || num_envs <= 0) return fail(RR_EINVAL, "rr_batch_create: bad argument");
  if (!pick_kernel(m)) return fail(RR_EUNSUPPORTED, "rr_batch_create: no kernel instance for this model's slot counts");
  HIPCHK(hipSetDevice(device));
  rr_batch* b = new rr_batch();
  b->m = m; b->N = num_envs; b->device = device; b->stream = (hipStream_t)stream; b->kd = m->kd;
  int rc = 0;
#define UP(field, name) if ((rc = upload(b, name, &b->T.field))) { rr_batch_destroy(b); return rc; }
  UP(body_i, "k_body_i") UP(jnt_i, "k_jnt_i") UP(dof_i, "k_dof_i") UP(M_ij_k, "k_M_ij_k")
  UP(body_anc, "k_body_anc") UP(con_chain_rows, "k_con_chain_rows") UP(coljob, "k_coljob") UP(rowjob, "k_rowjob") UP(jobown, "k_jobown") UP(con_i, "k_con_i")
  UP(body_f, "k_body_f") UP(jnt_f, "k_jnt_f") UP(dof_f, "k_dof_f")
  UP(act_f, "k_act_f") UP(con_f, "k_con_f") UP(root_mass, "k_root_mass")
  UP(act_i, "k_act_i") UP(act_m_i, "k_act_m_i") UP(act_m_f, "k_act_m_f")
#undef UP
  {
    const uint32_t qb = (uint32_t)m->kd.o_qLD * 4u, ab = (uint32_t)m->kd.o_xpos * 4u, nM = (uint32_t)m->dims.nM;
    if ((rc = upload_levels(b, m->solver == 2 ? "k_factor3p" : "k_factor3", &b->T.factor3, qb, ab, nM)) || (rc = upload_levels(b, "k_linv", &b->T.linv, qb, ab, nM))) { rr_batch_destroy(b); return rc; }
  }
  memset(&b->T_rep, 0, sizeof(b->T_rep));
  if (m->pair_ok) {      // tables of one replica; the level schedules twice (second copy addressed into the second wavefront's region)
#define UPH(field, name) if ((rc = upload(b, "h_" name, &b->T_rep.field))) { rr_batch_destroy(b); return rc; }
    UPH(body_i, "k_body_i") UPH(jnt_i, "k_jnt_i") UPH(dof_i, "k_dof_i") UPH(M_ij_k, "k_M_ij_k")
    UPH(body_anc, "k_body_anc") UPH(con_chain_rows, "k_con_chain_rows") UPH(coljob, "k_coljob") UPH(rowjob, "k_rowjob") UPH(jobown, "k_jobown") UPH(con_i, "k_con_i")
    UPH(body_f, "k_body_f") UPH(jnt_f, "k_jnt_f") UPH(dof_f, "k_dof_f")
    UPH(act_f, "k_act_f") UPH(con_f, "k_con_f") UPH(root_mass, "k_root_mass")
#undef UPH
    const uint32_t base = (uint32_t)m->kd_rep.o_qLD * 4u, rb = (uint32_t)m->kd_rep.lds_bytes_rep;
    const uint32_t abase = (uint32_t)m->kd_rep.o_xpos * 4u, hnM = (uint32_t)m->kd_rep.nM;
    if ((rc = upload_levels(b, "h_k_factor3", &b->T_rep.factor3, base, abase, hnM, 2, rb)) || (rc = upload_levels(b, "h_k_linv", &b->T_rep.linv, base, abase, hnM, 2, rb))) { rr_batch_destroy(b); return rc; }
    b->T_rep.anc4 = b->T_rep.M_ij_k;     // unused (Newton only); a valid pointer
  }
  {   // ancestor ids along the rows of M (Newton): byte Madr[i] + p = the p-th ancestor of dof i (p = 0: i itself)
    const Entry *an = m->find("dof_anc"), *ad = m->find("dof_ancadr"), *ma = m->find("dof_Madr");
    std::vector<unsigned char> bytes(((size_t)m->dims.nM + 3) / 4 * 4 + 4, 0);
    if (an && ad && ma) {
      const int32_t *anc = (const int32_t*)an->data, *adr = (const int32_t*)ad->data, *madr = (const int32_t*)ma->data;
      for (int i = 0; i < m->dims.nv; ++i) {
        const int n = adr[i + 1] - adr[i];                 // chain root .. self
        for (int p = 0; p < n; ++p) bytes[madr[i] + p] = (unsigned char)anc[adr[i] + n - 1 - p];
      }
    } else if (m->solver == 2) { rr_batch_destroy(b); return fail(RR_EIO, "rr_batch_create: blob lacks the ancestor tables the Newton solver needs"); }
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, bytes.size()));
    b->dev_allocs.push_back(p);
    HIPCHK(hipMemcpy(p, bytes.data(), bytes.size(), hipMemcpyHostToDevice));
    b->T.anc4 = (rr_gi)p;
  }
  kern_t kern = pick_kernel(m);
  if (!m->stage_ok) { rr_batch_destroy(b); return fail(RR_EUNSUPPORTED, "rr_batch_create: 4*ncon + nv exceeds the line-search staging cells (6*nbody)"); }
  if (m->dims.lds_bytes > 64 * 1024) { rr_batch_destroy(b); return fail(RR_EUNSUPPORTED, "rr_batch_create: per-env working set exceeds the 64 KiB of LDS one workgroup may address"); }
  kern_t pair_kern = m->pair_ok ? pick_pair_kernel(m) : nullptr;
  for (kern_t kk : {kern, pick_kernel(m, false, true), pick_unroll_kernel(m), pick_unroll_kernel(m, true), pick_kernel(m, true), pair_kern}) {   // every instance a launch may pick
    if (!kk) continue;
    hipError_t e = hipFuncSetAttribute((const void*)kk, hipFuncAttributeMaxDynamicSharedMemorySize, kk == pair_kern ? 2 * m->kd_rep.lds_bytes_rep + 128 : m->dims.lds_bytes);
    if (e != hipSuccess) { rr_batch_destroy(b); return fail(RR_EHIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); }
    // the level schedules address LDS by absolute byte address, so the dynamic segment must begin at LDS address 0: no static LDS
    hipFuncAttributes fa;
    e = hipFuncGetAttributes(&fa, (const void*)kk);
    if (e != hipSuccess) { rr_batch_destroy(b); return fail(RR_EHIP, std::string("hipFuncGetAttributes: ") + hipGetErrorString(e)); }
    if (fa.sharedSizeBytes != 0) { rr_batch_destroy(b); return fail(RR_EUNSUPPORTED, "rr_batch_create: a step-kernel instance has static LDS (the level schedules need the dynamic segment at address 0)"); }
  }
  if (m->dyn) {
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, 64));
    HIPCHK(hipMemset(p, 0, 64));
    b->dev_allocs.push_back(p);
    b->dyn_overflow = (unsigned*)p;
  }
  {
    const char* pace = getenv("RR_PACE");
    if (!(pace && pace[0] == '0')) {
      void* p = nullptr;
      HIPCHK(hipMalloc(&p, 64));
      b->dev_allocs.push_back(p);
      b->progress = (unsigned*)p;
    }
  }
  m->live_batches.fetch_add(1);
  b->counted = true;
  *out = b;
  return RR_OK;
}

extern "C" void rr_batch_destroy(rr_batch* b) {
  if (!b) return;
  if (b->counted) b->m->live_batches.fetch_sub(1);
  for (void* p : b->dev_allocs) (void)hipFree(p);
  for (hipEvent_t e : b->ev0) (void)hipEventDestroy(e);
  for (hipEvent_t e : b->ev1) (void)hipEventDestroy(e);
  delete b;
}

#define RR_TIMING_RING 256
// Fold finished event pairs into the totals.  `all`: wait for every pending launch (rr_batch_kernel_time).  Otherwise (the ring is
// full at a launch) only pairs that HAVE finished are taken, oldest first, without blocking -- a blocking drain stalled the host
// for the length of the launch in flight, long enough with multi-step launches for a second stream to run dry -- and if none has,
// the oldest one is waited for.
static int collect_timing(rr_batch* b, bool all = true) {
  bool first = true;
  while (b->npending > 0) {
    const int i = b->ring_head;
    if (all || (first && hipEventQuery(b->ev1[i]) != hipSuccess)) HIPCHK(hipEventSynchronize(b->ev1[i]));
    else if (hipEventQuery(b->ev1[i]) != hipSuccess) break;
    first = false;
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, b->ev0[i], b->ev1[i]));
    b->total_ms += ms;
    b->launches += 1;
    b->ring_head = (i + 1) % RR_TIMING_RING;
    b->npending -= 1;
  }
  (void)hipGetLastError();      // hipEventQuery reports "not ready" through the sticky last-error slot
  return RR_OK;
}

static int launch(rr_batch* b, const rr_state* st, const float* ctrl, int n_frames, const rr_env_io* env, const rr_outputs* out, int mode,
                  const rr_state* st_in = nullptr, const int32_t* cur_frame_in = nullptr, const rr_unroll_io* un = nullptr, int unroll_T = 0,
                  const rr_actor_io* ac = nullptr) {
  if (!b || !st || !st->qpos || !st->qvel || !st->act || !st->qacc_warmstart) return fail(RR_EINVAL, "launch: null state pointer");
  if (st_in && (!st_in->qpos || !st_in->qvel || !st_in->act || !st_in->qacc_warmstart)) return fail(RR_EINVAL, "launch: null input state pointer");
  if ((mode & 1) && (!ctrl || n_frames <= 0)) return fail(RR_EINVAL, "launch: step needs ctrl and n_frames > 0");
  RRIO io;
  memset(&io, 0, sizeof(io));
  io.qpos = st->qpos; io.qvel = st->qvel; io.act = st->act; io.warm = st->qacc_warmstart; io.ctrl = ctrl;
  const rr_state* si = st_in ? st_in : st;
  io.qpos_in = si->qpos; io.qvel_in = si->qvel; io.act_in = si->act; io.warm_in = si->qacc_warmstart;
  if (out) {
    io.o_cinert = out->cinert; io.o_cvel = out->cvel; io.o_qfrc_actuator = out->qfrc_actuator; io.o_xpos = out->xpos;
    io.o_xmat = out->xmat; io.o_com = out->subtree_com; io.dbg = out->debug;
    io.o_cdist = out->contact_dist; io.o_cpos = out->contact_pos; io.o_cframe = out->contact_frame;
  }
  if (env) {
    if (!env->obs || !env->track_pos || !env->cur_frame || env->track_len <= 0) return fail(RR_EINVAL, "launch: env io needs obs, track_pos, cur_frame");
    if ((mode & 1) && (!env->reward || !env->done || !env->metrics)) return fail(RR_EINVAL, "launch: env step needs reward, done, metrics");
    io.track_pos = env->track_pos; io.track_len = env->track_len; io.cur_frame = env->cur_frame; io.obs = env->obs;
    io.cur_frame_in = cur_frame_in ? cur_frame_in : env->cur_frame;
    io.reward = env->reward; io.done = env->done; io.metrics = env->metrics;
    io.healthy_reward = env->healthy_reward; io.ctrl_cost_weight = env->ctrl_cost_weight; io.z_min = env->healthy_z_min;
    io.z_max = env->healthy_z_max; io.terminate_when_unhealthy = env->terminate_when_unhealthy;
  }
  io.mode = mode;
  HIPCHK(hipSetDevice(b->device));
  kern_t kern = pick_kernel(b->m, b->prof != nullptr, io.dbg != nullptr || io.o_cdist || io.o_cpos || io.o_cframe);
  if (un) {      // multi-step rollout: the UNROLL instance, no diagnostics
    if (b->prof || io.dbg || io.o_cdist || io.o_cpos || io.o_cframe || out) return fail(RR_EUNSUPPORTED, "rr_env_unroll: no diagnostic outputs in a multi-step rollout");
    kern = pick_unroll_kernel(b->m, ac != nullptr);
    if (!kern) return fail(RR_EUNSUPPORTED, "rr_env_unroll: no multi-step kernel instance for this model / solver");
    if (ac) {
      if (b->m->dims.obs_dim > 1280) return fail(RR_EUNSUPPORTED, "rr_env_unroll_policy: observation wider than 1280");
      if (ac->nhidden < 1 || ac->nhidden > 5) return fail(RR_EUNSUPPORTED, "rr_env_unroll_policy: 1 .. 5 hidden layers");
      io.a_obs_in = ac->obs_in; io.a_mean = ac->mean; io.a_std = ac->std; io.a_W0 = ac->w0; io.a_b0 = ac->b0;
      for (int l = 1; l < ac->nhidden; ++l) { io.a_Wt[l - 1] = ac->hidden_wt[l - 1]; io.a_b[l - 1] = ac->hidden_b[l - 1]; }
      io.a_Wth = ac->head_wt; io.a_bh = ac->head_b; io.a_noise = ac->noise; io.a_actions = ac->actions_out; io.ctrl = ac->actions_out;
      io.t_obs = ac->traj_obs; io.t_raw = ac->traj_raw_action; io.t_logp = ac->traj_log_prob; io.t_reward = ac->traj_reward;
      io.t_discount = ac->traj_discount; io.t_trunc = ac->traj_truncation; io.a_min_std = ac->min_std; io.a_nh = ac->nhidden;
      io.a_seg = ac->segment_length > 0 ? ac->segment_length : unroll_T;
      if (unroll_T % io.a_seg) return fail(RR_EINVAL, "rr_env_unroll_policy: num_steps must be a multiple of segment_length");
      io.obs = ac->traj_obs;
    }
    io.first_qpos = un->first.qpos; io.first_qvel = un->first.qvel; io.first_act = un->first.act; io.first_warm = un->first.qacc_warmstart;
    io.first_obs = un->first_obs; io.prev_done = un->prev_done; io.steps_in = un->steps_in; io.steps_out = un->steps_out;
    io.trunc_out = un->truncation_out; io.episode_length = un->episode_length; io.unroll_T = unroll_T;
    if (b->progress && unroll_T > 1) {
      HIPCHK(hipMemsetAsync(b->progress, 0, 4, b->stream));
      io.progress = b->progress;
      static const struct Pace { float t[3]; int mode; } pace = [] {        // RR_PACE_T="t1,t2,t3", RR_PACE_MODE: tuning switches (tools/)
        Pace p = {{0.3f, 0.6f, 1.0f}, 0};
        if (const char* e = getenv("RR_PACE_T")) sscanf(e, "%f,%f,%f", &p.t[0], &p.t[1], &p.t[2]);
        if (const char* e = getenv("RR_PACE_MODE")) p.mode = atoi(e);
        return p;
      }();
      io.pace_t1 = pace.t[0]; io.pace_t2 = pace.t[1]; io.pace_t3 = pace.t[2]; io.pace_mode = pace.mode;
    }
  }
  if (!kern) return fail(RR_EUNSUPPORTED, "launch: no diagnostic kernel instance for this model");
  io.prof = b->prof;
  io.env_map = b->env_map; io.cost = b->cost;
  io.dyn_overflow = b->dyn_overflow;
  RRDims kd = b->kd;
  kd.iterations = b->m->kd.iterations; kd.ls_iterations = b->m->kd.ls_iterations;
  // two-tree model, physics only, no diagnostics: one wavefront per replica (rr_kernel.h PAIR)
  const bool pair = b->m->pair_ok && !env && !out && !un && !b->prof && !b->env_map && !b->cost && b->m->solver != 2;
  if (pair) {
    kern = pick_pair_kernel(b->m);
    kd = b->m->kd_rep;
    kd.iterations = b->m->kd.iterations; kd.ls_iterations = b->m->kd.ls_iterations;
  }
  if (b->timing) {
    if (b->npending == RR_TIMING_RING) { int rc = collect_timing(b, false); if (rc) return rc; }
    HIPCHK(hipEventRecord(b->ev0[(b->ring_head + b->npending) % RR_TIMING_RING], b->stream));
  }
  if (pair) hipLaunchKernelGGL(kern, dim3(b->N), dim3(2 * RR_LANES), (size_t)(2 * kd.lds_bytes_rep + 128), b->stream, kd, b->T_rep, io, b->N, n_frames);
  else hipLaunchKernelGGL(kern, dim3(b->N), dim3(RR_LANES), (size_t)b->m->dims.lds_bytes, b->stream, kd, b->T, io, b->N, n_frames);
  HIPCHK(hipGetLastError());
  if (b->timing) {
    HIPCHK(hipEventRecord(b->ev1[(b->ring_head + b->npending) % RR_TIMING_RING], b->stream));
    b->npending += 1;
  }
  return RR_OK;
}

extern "C" int rr_pipeline_step_to(rr_batch* b, const rr_state* in, const rr_state* outst, const float* ctrl, int32_t n_frames, const rr_outputs* out) {
  if (!in) return fail(RR_EINVAL, "rr_pipeline_step_to: null input state");
  return launch(b, outst, ctrl, n_frames, nullptr, out, 1, in);
}
extern "C" int rr_env_step_to(rr_batch* b, const rr_state* in, const rr_state* outst, const float* action, int32_t n_frames, const rr_env_io* env,
                              const int32_t* cur_frame_in, const rr_outputs* out) {
  if (!env) return fail(RR_EINVAL, "rr_env_step_to: env io required");
  if (!in || !cur_frame_in) return fail(RR_EINVAL, "rr_env_step_to: null input state");
  return launch(b, outst, action, n_frames, env, out, 1, in, cur_frame_in);
}
extern "C" int rr_batch_contact_overflow(rr_batch* b, int64_t* events) {
  if (!b || !events) return fail(RR_EINVAL, "rr_batch_contact_overflow: null argument");
  *events = 0;
  if (!b->dyn_overflow) return RR_OK;                // static-slot models hold every contact of the model: nothing can overflow
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  unsigned v = 0;
  HIPCHK(hipMemcpy(&v, b->dyn_overflow, sizeof(v), hipMemcpyDeviceToHost));
  *events = (int64_t)v;
  return RR_OK;
}
extern "C" int rr_batch_unroll_supported(const rr_batch* b, int32_t with_actor) {
  if (!b) return fail(RR_EINVAL, "rr_batch_unroll_supported: null batch");
  return pick_unroll_kernel(b->m, with_actor != 0) ? 1 : 0;
}
extern "C" int rr_env_unroll(rr_batch* b, const rr_state* in, const rr_state* outst, const float* actions, int32_t num_steps, int32_t n_frames,
                             const rr_env_io* env, const int32_t* cur_frame_in, const rr_unroll_io* wrap) {
  if (!env || !in || !cur_frame_in || !wrap || !actions || num_steps <= 0) return fail(RR_EINVAL, "rr_env_unroll: bad argument");
  if (!wrap->first.qpos || !wrap->first.qvel || !wrap->first.act || !wrap->first.qacc_warmstart || !wrap->first_obs || !wrap->prev_done ||
      !wrap->steps_in || !wrap->steps_out || !wrap->truncation_out)
    return fail(RR_EINVAL, "rr_env_unroll: null wrapper pointer");
  return launch(b, outst, actions, n_frames, env, nullptr, 1, in, cur_frame_in, wrap, num_steps);
}
extern "C" int rr_env_unroll_policy(rr_batch* b, const rr_state* in, const rr_state* outst, int32_t num_steps, int32_t n_frames, const rr_env_io* env,
                                    const int32_t* cur_frame_in, const rr_unroll_io* wrap, const rr_actor_io* actor) {
  if (!env || !in || !cur_frame_in || !wrap || !actor || num_steps <= 0) return fail(RR_EINVAL, "rr_env_unroll_policy: bad argument");
  if (!wrap->first.qpos || !wrap->first.qvel || !wrap->first.act || !wrap->first.qacc_warmstart || !wrap->first_obs || !wrap->prev_done ||
      !wrap->steps_in || !wrap->steps_out || !wrap->truncation_out)
    return fail(RR_EINVAL, "rr_env_unroll_policy: null wrapper pointer");
  if (!actor->obs_in || !actor->w0 || !actor->b0 || !actor->head_wt || !actor->head_b || !actor->noise || !actor->actions_out || !actor->traj_obs ||
      !actor->traj_raw_action || !actor->traj_log_prob || !actor->traj_reward || !actor->traj_discount || !actor->traj_truncation ||
      (actor->mean == nullptr) != (actor->std == nullptr))
    return fail(RR_EINVAL, "rr_env_unroll_policy: null actor pointer");
  for (int l = 1; l < actor->nhidden && l < 5; ++l)
    if (!actor->hidden_wt[l - 1] || !actor->hidden_b[l - 1]) return fail(RR_EINVAL, "rr_env_unroll_policy: null hidden layer");
  rr_env_io e = *env;
  e.obs = actor->traj_obs;                   // the observations of the launch go to the trajectory
  return launch(b, outst, actor->actions_out, n_frames, &e, nullptr, 1, in, cur_frame_in, wrap, num_steps, actor);
}
extern "C" int rr_pipeline_init(rr_batch* b, const rr_state* st, const rr_outputs* out) { return launch(b, st, nullptr, 1, nullptr, out, 0); }
extern "C" int rr_pipeline_step(rr_batch* b, const rr_state* st, const float* ctrl, int32_t n_frames, const rr_outputs* out) {
  return launch(b, st, ctrl, n_frames, nullptr, out, 1);
}
extern "C" int rr_env_step(rr_batch* b, const rr_state* st, const float* action, int32_t n_frames, const rr_env_io* env, const rr_outputs* out) {
  if (!env) return fail(RR_EINVAL, "rr_env_step: env io required");
  return launch(b, st, action, n_frames, env, out, 1);
}
extern "C" int rr_env_reset(rr_batch* b, const rr_state* st, const rr_env_io* env, const rr_outputs* out) {
  if (!env) return fail(RR_EINVAL, "rr_env_reset: env io required");
  return launch(b, st, nullptr, 1, env, out, 2);
}

// ------------------------------------------------------------------------------------------ PPO: GAE
__global__ void rr_gae_kernel(const float* __restrict__ trunc, const float* __restrict__ term, const float* __restrict__ rew,
                              const float* __restrict__ val, const float* __restrict__ boot, int T, int B, float lam, float disc,
                              float* __restrict__ vs, float* __restrict__ adv) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float acc = 0.0f;
  const float vb = boot[b];
  for (int t = T - 1; t >= 0; --t) {     // vs_t - v_t = delta_t + gamma (1-term_t)(1-trunc_t) lambda (vs_{t+1} - v_{t+1})
    const size_t i = (size_t)t * B + b;
    const float mask = 1.0f - trunc[i], nt = 1.0f - term[i];
    const float vnext = t == T - 1 ? vb : val[i + B];
    const float delta = (rew[i] + disc * nt * vnext - val[i]) * mask;
    acc = delta + disc * nt * mask * lam * acc;
    vs[i] = acc + val[i];
  }
  for (int t = 0; t < T; ++t) {          // advantages use vs_{t+1} (bootstrap at the end)
    const size_t i = (size_t)t * B + b;
    const float vsn = t == T - 1 ? vb : vs[i + B];
    adv[i] = (rew[i] + disc * (1.0f - term[i]) * vsn - val[i]) * (1.0f - trunc[i]);
  }
}

extern "C" int rr_compute_gae(const float* truncation, const float* termination, const float* rewards, const float* values,
                              const float* bootstrap_value, int32_t T, int32_t B, float lambda_, float discount, float* vs,
                              float* advantages, void* stream) {
  if (!truncation || !termination || !rewards || !values || !bootstrap_value || !vs || !advantages || T <= 0 || B <= 0)
    return fail(RR_EINVAL, "rr_compute_gae: bad argument");
  hipLaunchKernelGGL(rr_gae_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, truncation, termination, rewards,
                     values, bootstrap_value, T, B, lambda_, discount, vs, advantages);
  HIPCHK(hipGetLastError());
  return RR_OK;
}

// ------------------------------------------------------------------------------------------ PPO: loss + gradient w.r.t. the network outputs
static void ppo_blocks(int T, int B, int* nblk1, int* nblk2) {
  *nblk1 = (B + 255) / 256;
  const long n = (long)T * B;
  *nblk2 = (int)std::min<long>((n + 7) / 8, 1024);
}
extern "C" size_t rr_ppo_loss_workspace_bytes(int32_t T, int32_t B) {
  if (T <= 0 || B <= 0) return 0;
  int n1, n2;
  ppo_blocks(T, B, &n1, &n2);
  return rr_align_up((size_t)2 * T * B * sizeof(float), 8) + ((size_t)2 * n1 + (size_t)3 * n2) * sizeof(double);
}
extern "C" int rr_ppo_loss(const float* policy_logits, const float* values, const float* raw_action, const float* log_prob, const float* reward,
                           const float* discount, const float* truncation, const int64_t* idx, const float* noise, int32_t T, int32_t B, int32_t A,
                           const rr_ppo_cfg* cfg, float* grad_logits, float* grad_values, float* metrics, void* workspace, size_t workspace_bytes,
                           void* stream) {
  if (!policy_logits || !values || !raw_action || !log_prob || !reward || !discount || !truncation || !noise || !cfg || !grad_logits ||
      !grad_values || !metrics || !workspace || T <= 0 || B <= 0 || A <= 0)
    return fail(RR_EINVAL, "rr_ppo_loss: bad argument");
  if (workspace_bytes < rr_ppo_loss_workspace_bytes(T, B) || ((uintptr_t)workspace & 7))
    return fail(RR_EINVAL, "rr_ppo_loss: workspace too small (rr_ppo_loss_workspace_bytes) or not 8-byte aligned");
  RRPpoArgs P;
  memset(&P, 0, sizeof(P));
  P.logits = policy_logits; P.values = values; P.raw_action = raw_action; P.log_prob = log_prob; P.reward = reward; P.discount = discount;
  P.truncation = truncation; P.idx = idx; P.noise = noise; P.T = T; P.B = B; P.A = A;
  P.entropy_cost = cfg->entropy_cost; P.discounting = cfg->discounting; P.reward_scaling = cfg->reward_scaling; P.gae_lambda = cfg->gae_lambda;
  P.clipping_epsilon = cfg->clipping_epsilon; P.min_std = cfg->min_std; P.normalize_advantage = cfg->normalize_advantage;
  P.grad_logits = grad_logits; P.grad_values = grad_values; P.metrics = metrics;
  ppo_blocks(T, B, &P.nblk1, &P.nblk2);
  char* w = (char*)workspace;
  P.vs = (float*)w; P.adv = P.vs + (size_t)T * B;
  P.part_adv = (double*)(w + rr_align_up((size_t)2 * T * B * sizeof(float), 8));
  P.part_loss = P.part_adv + 2 * P.nblk1;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(rr_ppo_gae_kernel, dim3(P.nblk1), dim3(256), 0, st, P);
  hipLaunchKernelGGL(rr_ppo_loss_kernel, dim3(P.nblk2), dim3(256), 0, st, P);
  hipLaunchKernelGGL(rr_ppo_metrics_kernel, dim3(1), dim3(64), 0, st, P);
  HIPCHK(hipGetLastError());
  return RR_OK;
}

extern "C" int rr_policy_sample(const float* logits, const float* noise, int32_t N, int32_t A, float min_std, float* action, float* raw_action,
                                float* log_prob, void* stream) {
  if (!logits || !noise || !action || !raw_action || !log_prob || N <= 0 || A <= 0) return fail(RR_EINVAL, "rr_policy_sample: bad argument");
  hipLaunchKernelGGL(rr_policy_sample_kernel, dim3((N + 7) / 8), dim3(256), 0, (hipStream_t)stream, logits, noise, N, A, min_std, action,
                     raw_action, log_prob);
  HIPCHK(hipGetLastError());
  return RR_OK;
}

// policy network backward: the delta chain of the 32-wide stack in one launch (csrc/rr_ppo.h)
static int pol_bwd_blocks(int M) { return std::max(1, std::min(1024, (M + 7) / 8)); }     // latency-bound per row: many small blocks (256 blocks: 76 us, 1024: 35 us)
extern "C" size_t rr_policy_backward_workspace_bytes(int32_t M, int32_t nhidden) {
  if (M <= 0 || nhidden <= 0) return 0;
  return (size_t)nhidden * pol_bwd_blocks(M) * 32 * sizeof(float);
}
extern "C" int rr_policy_backward(const float* grad_logits, const float* head_weight, const float* const* hidden_weights, int32_t nhidden, int32_t M,
                                  int32_t P, float* pre_act, int32_t pre_act_rows, float* delta, float* const* bias_grads, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  if (!grad_logits || !head_weight || !hidden_weights || !pre_act || !delta || !bias_grads || !workspace || M <= 0 || pre_act_rows < M)
    return fail(RR_EINVAL, "rr_policy_backward: bad argument");
  if (nhidden < 1 || nhidden > RR_POL_MAXL || P < 1 || P > 64) return fail(RR_EUNSUPPORTED, "rr_policy_backward: unsupported network shape");
  if (workspace_bytes < rr_policy_backward_workspace_bytes(M, nhidden)) return fail(RR_EINVAL, "rr_policy_backward: workspace too small");
  RRPolBwdArgs A;
  memset(&A, 0, sizeof(A));
  A.g = grad_logits; A.w_head = head_weight; A.z = pre_act; A.delta = delta; A.part = (float*)workspace; A.M = M; A.P = P; A.nh = nhidden; A.zrows = pre_act_rows;
  A.nblk = pol_bwd_blocks(M);
  for (int j = 0; j < nhidden; ++j) {
    if (!bias_grads[j] || (j > 0 && !hidden_weights[j])) return fail(RR_EINVAL, "rr_policy_backward: null layer pointer");
    A.bgrad[j] = bias_grads[j];
    A.W[j] = j > 0 ? hidden_weights[j] : nullptr;
  }
  const size_t lds = ((size_t)P * 32 + (size_t)(nhidden - 1) * 1024) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(rr_policy_backward_kernel, dim3(A.nblk), dim3(256), lds, st, A);
  hipLaunchKernelGGL(rr_policy_colsum_kernel, dim3(nhidden), dim3(1024), 0, st, A);
  HIPCHK(hipGetLastError());
  return RR_OK;
}

// elementwise half of a hidden SiLU layer's backward (csrc/rr_ppo.h)
static int silu_bwd_blocks(int M, int* rows_per_block) {
  const int target = 512;                                   // blocks: two per CU
  *rows_per_block = std::max(8, (M + target - 1) / target);
  return (M + *rows_per_block - 1) / *rows_per_block;
}
extern "C" size_t rr_mlp_silu_backward_workspace_bytes(int32_t M, int32_t H) {
  if (M <= 0 || H <= 0) return 0;
  int rpb;
  return (size_t)silu_bwd_blocks(M, &rpb) * H * sizeof(float);
}
extern "C" int rr_mlp_silu_backward(const float* g, const float* z, int32_t M, int32_t H, float* delta, float* h, float* bias_grad,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  if (!g || !z || !delta || !h || !bias_grad || !workspace || M <= 0 || H <= 0) return fail(RR_EINVAL, "rr_mlp_silu_backward: bad argument");
  if (H > 256 || 256 % H) return fail(RR_EUNSUPPORTED, "rr_mlp_silu_backward: the layer width must divide 256");
  if (workspace_bytes < rr_mlp_silu_backward_workspace_bytes(M, H)) return fail(RR_EINVAL, "rr_mlp_silu_backward: workspace too small");
  int rpb;
  const int nblk = silu_bwd_blocks(M, &rpb);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(rr_silu_bwd_kernel, dim3(nblk), dim3(256), 0, st, g, z, M, H, rpb, delta, h, (float*)workspace);
  hipLaunchKernelGGL(rr_colsum_kernel, dim3((H + 15) / 16), dim3(256), 0, st, (const float*)workspace, nblk, H, bias_grad);
  HIPCHK(hipGetLastError());
  return RR_OK;
}

// ------------------------------------------------------------------------------------------ PPO: fused MLP forward (MFMA f32)
static int mlp_net(const rr_mlp_net* n, int K, int hidden, bool is_value, RRMlpNet* out, const char* who) {
  memset(out, 0, sizeof(*out));
  if (!n) return RR_OK;
  if (!n->weights || !n->biases || !n->sizes || n->nlayers < 2 || n->nlayers > RR_MLP_MAXL) return fail(RR_EINVAL, std::string("rr_mlp_forward: bad ") + who + " network description");
  if (n->sizes[0] != K) return fail(RR_EINVAL, std::string("rr_mlp_forward: ") + who + " input width differs from the observation width");
  for (int l = 1; l < n->nlayers; ++l)
    if (n->sizes[l] != hidden) return fail(RR_EUNSUPPORTED, std::string("rr_mlp_forward: ") + who + " hidden width must be " + std::to_string(hidden));
  const int od = n->sizes[n->nlayers];
  if (is_value ? od != 1 : (od < 1 || od > 64)) return fail(RR_EUNSUPPORTED, std::string("rr_mlp_forward: unsupported ") + who + " output width");
  for (int l = 0; l < n->nlayers; ++l) {
    if (!n->weights[l] || !n->biases[l]) return fail(RR_EINVAL, std::string("rr_mlp_forward: null ") + who + " parameter");
    out->W[l] = n->weights[l]; out->b[l] = n->biases[l];
  }
  out->nlayers = n->nlayers; out->out_dim = od;
  return RR_OK;
}

extern "C" int rr_mlp_forward(const float* obs, const int64_t* obs_rows, int32_t M, int32_t K, const float* mean, const float* std_, const rr_mlp_net* policy,
                              const rr_mlp_net* value, float* policy_out, float* value_out, float* policy_pre, float* value_pre, void* stream) {
  if (!obs || M <= 0 || K <= 0 || (!policy && !value)) return fail(RR_EINVAL, "rr_mlp_forward: bad argument");
  if ((mean == nullptr) != (std_ == nullptr)) return fail(RR_EINVAL, "rr_mlp_forward: mean and std must be given together");
  if ((policy && !policy_out) || (value && !value_out)) return fail(RR_EINVAL, "rr_mlp_forward: missing output buffer");
  RRMlpArgs A;
  memset(&A, 0, sizeof(A));
  int rc;
  if ((rc = mlp_net(policy, K, RR_MLP_PH, false, &A.pol, "policy")) || (rc = mlp_net(value, K, RR_MLP_VH, true, &A.val, "value"))) return rc;
  A.obs = obs; A.rows = obs_rows; A.M = M; A.K = K; A.mean = mean; A.std_ = std_;
  A.pol_out = policy_out; A.val_out = value_out; A.pol_act = policy ? policy_pre : nullptr; A.val_act = value ? value_pre : nullptr;
  const size_t lds = RR_MLP_LDS_FLOATS * sizeof(float);
  typedef void (*fwd_t)(const RRMlpArgs);
  const fwd_t kern = policy && value ? (fwd_t)rr_mlp_forward_kernel<true, true> : (value ? (fwd_t)rr_mlp_forward_kernel<true, false> : (fwd_t)rr_mlp_forward_kernel<false, true>);
  static bool attr_set = false;
  if (!attr_set) {
    for (fwd_t k_ : {(fwd_t)rr_mlp_forward_kernel<true, true>, (fwd_t)rr_mlp_forward_kernel<true, false>, (fwd_t)rr_mlp_forward_kernel<false, true>})
      HIPCHK(hipFuncSetAttribute((const void*)k_, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  // diagnostic (RR_MLP_PROF=<file>, eager calls only): shader-clock stamps per phase and workgroup, summarised into the file after every call
  static const char* prof_path = getenv("RR_MLP_PROF");
  static unsigned long long* prof_dev = nullptr;
  const int nwg = (M + RR_MLP_BM - 1) / RR_MLP_BM;
  if (prof_path && nwg <= 4096) {
    if (!prof_dev) HIPCHK(hipMalloc((void**)&prof_dev, 4096 * 16 * sizeof(unsigned long long)));
    HIPCHK(hipMemsetAsync(prof_dev, 0, (size_t)nwg * 16 * sizeof(unsigned long long), (hipStream_t)stream));
    A.prof = prof_dev;
  }
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, (hipStream_t)stream, A);
  HIPCHK(hipGetLastError());
  if (A.prof) {
    std::vector<unsigned long long> h((size_t)nwg * 16);
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    HIPCHK(hipMemcpy(h.data(), prof_dev, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (FILE* f = fopen(prof_path, "a")) {
      unsigned long long t0 = ~0ull, t1 = 0;
      double sum[16] = {0};
      for (int w = 0; w < nwg; ++w) {
        t0 = std::min(t0, h[(size_t)w * 16]); t1 = std::max(t1, h[(size_t)w * 16 + 12]);
        unsigned long long prev = h[(size_t)w * 16];
        for (int i = 1; i <= 12; ++i) { const unsigned long long v = h[(size_t)w * 16 + i]; if (v) { sum[i] += (double)(v - prev); prev = v; } }
      }
      fprintf(f, "M %d workgroups %d span %llu ticks; mean ticks per workgroup: layer1_loop %.0f layer1_store %.0f policy %.0f", M, nwg, t1 - t0, sum[1] / nwg, sum[2] / nwg, sum[3] / nwg);
      for (int l = 1; l <= 4; ++l) fprintf(f, " | hidden%d loop %.0f store %.0f", l, sum[2 + 2 * l] / nwg, sum[3 + 2 * l] / nwg);
      fprintf(f, " | head %.0f\n", sum[12] / nwg);
      fclose(f);
    }
  }
  return RR_OK;
}

// the rollout's actor in two launches: first policy layer split over k, then the remaining layers + the tanh-normal head
#define RR_POL_KSLICES 8
extern "C" size_t rr_policy_act_workspace_bytes(int32_t M) { return M > 0 ? (size_t)RR_POL_KSLICES * M * RR_MLP_PH * sizeof(float) : 0; }
extern "C" int rr_policy_act(const float* obs, const int64_t* obs_rows, int32_t M, int32_t K, const float* mean, const float* std_,
                             const rr_mlp_net* policy, const float* noise, float min_std, float* action, float* raw_action, float* log_prob,
                             float* logits, void* workspace, size_t workspace_bytes, void* stream) {
  if (!obs || !policy || !action || !workspace || M <= 0 || K <= 0) return fail(RR_EINVAL, "rr_policy_act: bad argument");
  if ((mean == nullptr) != (std_ == nullptr)) return fail(RR_EINVAL, "rr_policy_act: mean and std must be given together");
  RRMlpNet net;
  int rc = mlp_net(policy, K, RR_MLP_PH, false, &net, "policy");
  if (rc) return rc;
  const int nh = net.nlayers - 1, P = net.out_dim, A_ = P / 2;
  if ((P & 1) || A_ > 32 || nh > RR_POL_MAXL - 1) return fail(RR_EUNSUPPORTED, "rr_policy_act: the head must be 2 x action_size <= 64 wide");
  if (workspace_bytes < rr_policy_act_workspace_bytes(M)) return fail(RR_EINVAL, "rr_policy_act: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  RRPolL1Args L;
  memset(&L, 0, sizeof(L));
  L.obs = obs; L.rows = obs_rows; L.mean = mean; L.std_ = std_; L.W = net.W[0]; L.M = M; L.K = K; L.part = (float*)workspace;
  const int nchunk = (K + RR_MLP_KC - 1) / RR_MLP_KC;
  L.chunks_per_slice = (nchunk + RR_POL_KSLICES - 1) / RR_POL_KSLICES;
  const int nslice = (nchunk + L.chunks_per_slice - 1) / L.chunks_per_slice;
  hipLaunchKernelGGL(rr_policy_l1_kernel, dim3((M + RR_MLP_BM - 1) / RR_MLP_BM, nslice), dim3(256), 0, st, L);
  RRPolTailArgs T;
  memset(&T, 0, sizeof(T));
  T.part = (const float*)workspace; T.nslice = nslice; T.M = M; T.P = P; T.A = A_; T.nh = nh; T.noise = noise; T.min_std = min_std;
  T.action = action; T.raw = raw_action; T.logp = log_prob; T.logits = logits;
  for (int l = 1; l <= nh; ++l) T.W[l] = net.W[l];
  for (int l = 0; l <= nh; ++l) T.b[l] = net.b[l];
  const size_t lds = ((size_t)(nh - 1) * 1024 + 2048 + (size_t)nh * 32 + 64) * sizeof(float);
  hipLaunchKernelGGL(rr_policy_tail_kernel, dim3(std::max(1, std::min(256, (M + 7) / 8))), dim3(256), lds, st, T);
  HIPCHK(hipGetLastError());
  return RR_OK;
}

// value network backward: the delta chain on the matrix cores (csrc/rr_mlp.h)
extern "C" size_t rr_mlp_value_backward_workspace_bytes(int32_t M, int32_t nhidden) {
  if (M <= 0 || nhidden <= 0) return 0;
  return (size_t)nhidden * ((M + RR_MLP_BM - 1) / RR_MLP_BM) * RR_MLP_VH * sizeof(float);
}
extern "C" int rr_mlp_value_backward(const float* grad_value, const float* head_weight, const float* const* hidden_weights_t, int32_t nhidden,
                                     int32_t M, float* pre_act, float* delta, float* const* bias_grads, void* workspace, size_t workspace_bytes,
                                     void* stream) {
  if (!grad_value || !head_weight || !hidden_weights_t || !pre_act || !delta || !bias_grads || !workspace || M <= 0)
    return fail(RR_EINVAL, "rr_mlp_value_backward: bad argument");
  if (nhidden < 1 || nhidden > RR_MLP_MAXL) return fail(RR_EUNSUPPORTED, "rr_mlp_value_backward: unsupported number of hidden layers");
  if (workspace_bytes < rr_mlp_value_backward_workspace_bytes(M, nhidden)) return fail(RR_EINVAL, "rr_mlp_value_backward: workspace too small");
  RRMlpBwdArgs A;
  memset(&A, 0, sizeof(A));
  A.g = grad_value; A.w_head = head_weight; A.z = pre_act; A.delta = delta; A.part = (float*)workspace; A.M = M; A.nh = nhidden;
  A.nblk = (M + RR_MLP_BM - 1) / RR_MLP_BM;
  for (int j = 0; j < nhidden; ++j) {
    if (!bias_grads[j] || (j > 0 && !hidden_weights_t[j])) return fail(RR_EINVAL, "rr_mlp_value_backward: null layer pointer");
    A.bgrad[j] = bias_grads[j];
    A.Wt[j] = j > 0 ? hidden_weights_t[j] : nullptr;
  }
  const size_t lds = RR_MLP_BWD_LDS_FLOATS * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    HIPCHK(hipFuncSetAttribute((const void*)rr_mlp_value_backward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(rr_mlp_value_backward_kernel, dim3(A.nblk), dim3(256), lds, st, A);
  hipLaunchKernelGGL(rr_mlp_colsum_kernel, dim3(RR_MLP_VH / 16, nhidden), dim3(256), 0, st, A);
  HIPCHK(hipGetLastError());
  return RR_OK;
}

// weight gradient dW = delta' h as a split-row matrix-core product (csrc/rr_mlp.h)
#ifndef RR_DW_KC128
#define RR_DW_KC128 16      // rows of a staged chunk of the 128 x 128-tile weight-gradient kernel (32: half the LDS hand-offs, twice the bytes in flight)
#endif
struct DwPlan { int to, ti, kc, rows_per_slice, nslice; };
// `group_tiles`: output tiles of ALL products that share this product's launch (rr_mlp_weight_grad_batch runs the products of one tile
// shape as one launch, item = grid z); 0 = the product is launched alone.  The slices are cut so that the LAUNCH has ~target workgroups:
// planned per product (round 2), the four 256 x 256 hidden-layer products of a minibatch were cut into 128 slices each although their
// launch already holds 36 tiles -- 134 MB of partial tiles written and read back where 29 MB do.
static DwPlan dw_plan(int M, int O, int I, int group_tiles = 0) {
  DwPlan p;
  p.to = O <= 32 ? 32 : (O <= 64 ? 64 : 128);
  p.ti = O <= 32 ? 128 : (O <= 64 ? 64 : 128);
  p.kc = O <= 32 ? 64 : (O <= 64 ? 32 : RR_DW_KC128);    // equal matrix-core work per LDS hand-off in the three tile shapes
  const int tiles = group_tiles > 0 ? group_tiles : ((O + p.to - 1) / p.to) * ((I + p.ti - 1) / p.ti);
  static const int target = [] { const char* e = getenv("RR_DW_TARGET_WGS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 512; }();   // workgroups per product
  static const int target_batch = [] { const char* e = getenv("RR_DW_TARGET_WGS_BATCH"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 1024; }();   // ... per batched launch: two resident rounds
  const int want = std::max(1, std::min((group_tiles > 0 ? target_batch : target) / tiles, M / (2 * p.kc)));
  p.rows_per_slice = ((M + want - 1) / want + p.kc - 1) / p.kc * p.kc;
  p.nslice = (M + p.rows_per_slice - 1) / p.rows_per_slice;
  return p;
}
extern "C" size_t rr_mlp_weight_grad_batch_workspace_bytes(const rr_dw_item* items, int32_t n);
extern "C" size_t rr_mlp_weight_grad_workspace_bytes(int32_t M, int32_t O, int32_t I) {      // a single product = a batch of one
  if (M <= 0 || O <= 0 || I <= 0) return 0;
  rr_dw_item it;
  memset(&it, 0, sizeof(it));
  it.M = M; it.O = O; it.I = I;
  return rr_mlp_weight_grad_batch_workspace_bytes(&it, 1);
}
template <int GO, int GI, int WO, int WI, int KC>
static int dw_launch(const RRDwBatch& B, dim3 grid, hipStream_t st) {
  constexpr int TO = GO * WO * 32, TI = GI * WI * 32;
  constexpr size_t lds = (size_t)KC * ((TO + (TO % 64 == 0 ? 32 : 0)) + (TI + (TI % 64 == 0 ? 32 : 0))) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    HIPCHK(hipFuncSetAttribute((const void*)rr_mlp_dw_kernel<GO, GI, WO, WI, KC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL((rr_mlp_dw_kernel<GO, GI, WO, WI, KC>), grid, dim3(256), lds, st, B);
  return RR_OK;
}
// tiles of each tile-shape group of a batch (the products of one shape share a launch)
static void dw_group_tiles(const rr_dw_item* items, int n, int* gt) {
  gt[0] = gt[1] = gt[2] = 0;
  for (int i = 0; i < n; ++i) {
    const DwPlan p = dw_plan(items[i].M, items[i].O, items[i].I);
    gt[p.to == 32 ? 0 : (p.to == 64 ? 1 : 2)] += ((items[i].O + p.to - 1) / p.to) * ((items[i].I + p.ti - 1) / p.ti);
  }
}
static DwPlan dw_plan_batch(const rr_dw_item& it, const int* gt) {
  const DwPlan p = dw_plan(it.M, it.O, it.I);
  return dw_plan(it.M, it.O, it.I, gt[p.to == 32 ? 0 : (p.to == 64 ? 1 : 2)]);
}
extern "C" size_t rr_mlp_weight_grad_batch_workspace_bytes(const rr_dw_item* items, int32_t n) {
  size_t t = 0;
  int gt[3];
  if (!items || n <= 0) return 0;
  for (int i = 0; i < n; ++i) if (items[i].M <= 0 || items[i].O <= 0 || items[i].I <= 0) return 0;
  dw_group_tiles(items, n, gt);
  for (int i = 0; i < n; ++i) t += rr_align_up((size_t)dw_plan_batch(items[i], gt).nslice * items[i].O * items[i].I * sizeof(float), 256);
  return t;
}
extern "C" int rr_mlp_weight_grad_batch(const rr_dw_item* items, int32_t n, void* workspace, size_t workspace_bytes, void* stream) {
  if (!items || n <= 0 || !workspace) return fail(RR_EINVAL, "rr_mlp_weight_grad_batch: bad argument");
  if (n > RR_DW_MAXB) return fail(RR_EUNSUPPORTED, "rr_mlp_weight_grad_batch: at most 12 products per call");
  if (workspace_bytes < rr_mlp_weight_grad_batch_workspace_bytes(items, n)) return fail(RR_EINVAL, "rr_mlp_weight_grad_batch: workspace too small");
  RRDwBatch all, grp[3];
  memset(&all, 0, sizeof(all));
  memset(grp, 0, sizeof(grp));
  dim3 ggrid[3] = {dim3(0, 0, 0), dim3(0, 0, 0), dim3(0, 0, 0)};
  unsigned red_blocks = 0;
  char* w = (char*)workspace;
  int gt[3];
  for (int i = 0; i < n; ++i)
    if (!items[i].delta || !items[i].act || !items[i].grad || items[i].M <= 0 || items[i].O <= 0 || items[i].I <= 0) return fail(RR_EINVAL, "rr_mlp_weight_grad_batch: bad item");
  dw_group_tiles(items, n, gt);
  for (int i = 0; i < n; ++i) {
    const rr_dw_item& it = items[i];
    if ((it.mean == nullptr) != (it.std == nullptr) || (it.mean && !it.delta_colsum))
      return fail(RR_EINVAL, "rr_mlp_weight_grad_batch: mean, std and delta_colsum must be given together");
    const DwPlan p = dw_plan_batch(it, gt);
    RRDwArgs A;
    memset(&A, 0, sizeof(A));
    A.rows_per_slice = p.rows_per_slice; A.nslice = p.nslice;
    A.a = it.delta; A.b = it.act; A.rows = it.act_rows; A.mean = it.mean; A.std_ = it.std; A.bsum = it.delta_colsum; A.M = it.M; A.O = it.O; A.I = it.I;
    A.part = (float*)w; A.out = it.grad;
    w += rr_align_up((size_t)p.nslice * it.O * it.I * sizeof(float), 256);
    all.it[all.n++] = A;
    const int g = p.to == 32 ? 0 : (p.to == 64 ? 1 : 2);
    grp[g].it[grp[g].n++] = A;
    ggrid[g].x = std::max<unsigned>(ggrid[g].x, ((it.O + p.to - 1) / p.to) * ((it.I + p.ti - 1) / p.ti));
    ggrid[g].y = std::max<unsigned>(ggrid[g].y, (unsigned)p.nslice);
    ggrid[g].z = grp[g].n;
    red_blocks = std::max<unsigned>(red_blocks, (unsigned)(((size_t)it.O * it.I + 1023) / 1024));
  }
  hipStream_t st = (hipStream_t)stream;
  int rc = RR_OK;
  if (grp[2].n) rc = dw_launch<2, 2, 2, 2, RR_DW_KC128>(grp[2], ggrid[2], st);      // the big tiles first: they are the long ones
  if (!rc && grp[1].n) rc = dw_launch<2, 2, 1, 1, 32>(grp[1], ggrid[1], st);
  if (!rc && grp[0].n) rc = dw_launch<1, 4, 1, 1, 64>(grp[0], ggrid[0], st);
  if (rc) return rc;
  hipLaunchKernelGGL(rr_mlp_dw_reduce_kernel, dim3(red_blocks, all.n), dim3(256), 0, st, all);
  HIPCHK(hipGetLastError());
  return RR_OK;
}
extern "C" int rr_mlp_weight_grad(const float* delta, const float* act, const int64_t* act_rows, const float* mean, const float* std_,
                                  const float* delta_colsum, int32_t M, int32_t O, int32_t I, float* grad, void* workspace, size_t workspace_bytes,
                                  void* stream) {
  if (!delta || !act || !grad || !workspace || M <= 0 || O <= 0 || I <= 0) return fail(RR_EINVAL, "rr_mlp_weight_grad: bad argument");
  if (workspace_bytes < rr_mlp_weight_grad_workspace_bytes(M, O, I)) return fail(RR_EINVAL, "rr_mlp_weight_grad: workspace too small (rr_mlp_weight_grad_workspace_bytes)");
  rr_dw_item it;
  memset(&it, 0, sizeof(it));
  it.delta = delta; it.act = act; it.act_rows = act_rows; it.mean = mean; it.std = std_; it.delta_colsum = delta_colsum; it.M = M; it.O = O; it.I = I;
  it.grad = grad;
  return rr_mlp_weight_grad_batch(&it, 1, workspace, rr_align_up(workspace_bytes, 256), stream);
}

// ------------------------------------------------------------------------------------------ observation normaliser sums (csrc/rr_ppo.h)
static void mom_plan(long long nrows, int* rows_per_block, int* nblk) {
  const long long target = 2048;                              // blocks: one resident round of 8 per CU
  long long rpb = (nrows + target - 1) / target;
  if (rpb < 16) rpb = 16;
  *rows_per_block = (int)rpb;
  *nblk = (int)((nrows + rpb - 1) / rpb);
}
extern "C" size_t rr_obs_moments_workspace_bytes(int64_t nseq, int32_t T, int32_t K) {
  if (nseq <= 0 || T <= 0 || K <= 0) return 0;
  int rpb, nblk;
  mom_plan((long long)nseq * T, &rpb, &nblk);
  return (size_t)nblk * 2 * K * sizeof(double);
}
extern "C" int rr_obs_moments(const float* obs, int64_t nseq, int32_t Tp1, int32_t T, int32_t K, const float* mean, double* sums, void* workspace,
                              size_t workspace_bytes, void* stream) {
  if (!obs || !mean || !sums || !workspace || nseq <= 0 || T <= 0 || Tp1 < T || K <= 0) return fail(RR_EINVAL, "rr_obs_moments: bad argument");
  if (workspace_bytes < rr_obs_moments_workspace_bytes(nseq, T, K) || ((uintptr_t)workspace & 7) || ((uintptr_t)sums & 7))
    return fail(RR_EINVAL, "rr_obs_moments: workspace too small (rr_obs_moments_workspace_bytes) or not 8-byte aligned");
  RRMomArgs A;
  memset(&A, 0, sizeof(A));
  A.obs = obs; A.mean = mean; A.nrows = (long long)nseq * T; A.Tp1 = Tp1; A.T = T; A.K = K; A.part = (double*)workspace; A.out = sums;
  mom_plan(A.nrows, &A.rows_per_block, &A.nblk);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(rr_obs_moments_kernel, dim3(A.nblk), dim3(256), 0, st, A);
  hipLaunchKernelGGL(rr_obs_moments_reduce_kernel, dim3((2 * K + 255) / 256), dim3(256), 0, st, A);
  HIPCHK(hipGetLastError());
  return RR_OK;
}

// ------------------------------------------------------------------------------------------ training wrappers, fused
// brax.envs.wrappers.training: EpisodeWrapper (step count, truncation, done at episode end) followed by AutoResetWrapper
// (restore the stored first state where done) -- ~15 elementwise launches per env step when composed from tensor ops;
// here one launch, one 256-thread block per env.  Arrays are row-major [N][width]; `cur` is overwritten in place.
#define RR_WRAP_MAX 12
struct RRWrapArgs { const float* first[RR_WRAP_MAX]; float* cur[RR_WRAP_MAX]; int width[RR_WRAP_MAX]; int narr; };
__global__ void rr_wrap_kernel(const RRWrapArgs A, const float* __restrict__ prev_done, const float* __restrict__ prev_steps,
                               float* __restrict__ done, float* __restrict__ steps, float* __restrict__ truncation,
                               float episode_length, float action_repeat) {
  const int e = blockIdx.x;
  // AutoReset (before the step): steps <- 0 where the incoming state was done;  Episode: steps += action_repeat, done at the
  // episode end, truncation = over & !terminated;  AutoReset (after): first state where done
  const float s0 = prev_done[e] != 0.0f ? 0.0f : prev_steps[e];
  const float s1 = s0 + action_repeat;
  const bool over = s1 >= episode_length;
  const float d_env = done[e];
  const float d_out = over ? 1.0f : d_env;
  __syncthreads();                       // every thread has read done[e] before thread 0 rewrites it
  if (threadIdx.x == 0) { steps[e] = s1; truncation[e] = over ? 1.0f - d_env : 0.0f; done[e] = d_out; }
  if (d_out != 0.0f) {
    for (int a = 0; a < A.narr; ++a) {
      const int w = A.width[a];
      const float* src = A.first[a] + (size_t)e * w;
      float* dst = A.cur[a] + (size_t)e * w;
      for (int i = threadIdx.x; i < w; i += blockDim.x) dst[i] = src[i];
    }
  }
}

extern "C" int rr_wrap_episode_autoreset(int32_t num_envs, int32_t narr, const float* const* first, float* const* cur, const int32_t* widths,
                                         const float* prev_done, const float* prev_steps, float* done, float* steps, float* truncation,
                                         float episode_length, float action_repeat, void* stream) {
  if (num_envs <= 0 || narr < 0 || narr > RR_WRAP_MAX || !prev_done || !prev_steps || !done || !steps || !truncation)
    return fail(RR_EINVAL, "rr_wrap_episode_autoreset: bad argument");
  RRWrapArgs A;
  memset(&A, 0, sizeof(A));
  A.narr = narr;
  for (int i = 0; i < narr; ++i) {
    if (!first[i] || !cur[i] || widths[i] <= 0) return fail(RR_EINVAL, "rr_wrap_episode_autoreset: null array");
    A.first[i] = first[i]; A.cur[i] = cur[i]; A.width[i] = widths[i];
  }
  hipLaunchKernelGGL(rr_wrap_kernel, dim3(num_envs), dim3(256), 0, (hipStream_t)stream, A, prev_done, prev_steps, done, steps, truncation,
                     episode_length, action_repeat);
  HIPCHK(hipGetLastError());
  return RR_OK;
}

extern "C" int rr_debug_layout(const rr_batch* b, const char*** names, const int32_t** offsets, const int32_t** sizes) {
  if (!b) return fail(RR_EINVAL, "rr_debug_layout: null batch");
  if (names) *names = const_cast<const char**>(b->m->dbg_cnames.data());
  if (offsets) *offsets = b->m->dbg_off.data();
  if (sizes) *sizes = b->m->dbg_size.data();
  return (int)b->m->dbg_names.size();
}

extern "C" int rr_batch_set_profile(rr_batch* b, uint64_t* dev_cycles) {
  if (!b) return fail(RR_EINVAL, "rr_batch_set_profile: null batch");
  if (dev_cycles) {
    kern_t kern = pick_kernel(b->m, true);
    if (!kern) return fail(RR_EUNSUPPORTED, "rr_batch_set_profile: no diagnostic kernel instance for this model");
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, b->m->dims.lds_bytes);
    if (e != hipSuccess) return fail(RR_EHIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
  }
  b->prof = (unsigned long long*)dev_cycles;
  return RR_OK;
}

extern "C" int rr_batch_set_schedule(rr_batch* b, const int32_t* env_map, uint32_t* cost_cycles) {
  if (!b) return fail(RR_EINVAL, "rr_batch_set_schedule: null batch");
  b->env_map = env_map; b->cost = cost_cycles;
  return RR_OK;
}

extern "C" int rr_batch_set_timing(rr_batch* b, int32_t enable) {
  if (!b) return fail(RR_EINVAL, "rr_batch_set_timing: null batch");
  HIPCHK(hipSetDevice(b->device));
  if (enable && b->ev0.empty()) {
    for (int i = 0; i < RR_TIMING_RING; ++i) {
      hipEvent_t a, c;
      HIPCHK(hipEventCreate(&a)); b->ev0.push_back(a);
      HIPCHK(hipEventCreate(&c)); b->ev1.push_back(c);
    }
  }
  b->timing = enable != 0;
  b->total_ms = 0; b->launches = 0; b->npending = 0; b->ring_head = 0;
  return RR_OK;
}
extern "C" int rr_batch_kernel_time(rr_batch* b, double* total_ms, int64_t* launches) {
  if (!b) return fail(RR_EINVAL, "rr_batch_kernel_time: null batch");
  int rc = collect_timing(b);
  if (rc) return rc;
  if (total_ms) *total_ms = b->total_ms;
  if (launches) *launches = b->launches;
  return RR_OK;
}
