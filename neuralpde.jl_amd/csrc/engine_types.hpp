// engine_types.hpp — data model of the host side of the C ABI, shared by descriptor.cpp (parsing), program.cpp (residual-program
// passes), plan.cpp (kernel selection, buffers, reduction maps) and engine.cpp (evaluation + the extern "C" entry points).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/pinn_hip.h"
#include "aux_limits.hpp"
#include "plat.hpp"
#include "spec_registry.hpp"

namespace pe {

constexpr int REDUCE_SPLIT = 32;     // stage-1 chunks of the fixed-order slab reduction

extern thread_local std::string g_err;      // message of the last failed call on this thread (pinn_last_error)
int fail(const std::string& m);             // records the message, returns 1

constexpr int MAX_DERIV_ORDER = 6;
struct Slot {
    int net;
    int order;
    int axes[MAX_DERIV_ORDER];       // sorted; order <= 2 may be mixed in the fixed channel categories, anything else rides a general set
    unsigned lap = 0;        // != 0: the sum of the pure second derivatives over these axes (one "forward Laplacian" jet channel)
};
struct Net {
    int act;                         // ACT_TANH / ACT_SIGMOID / ACT_SIN on every hidden layer, or ACT_MIXED with
    int act_layers = 0;              //   the kind of hidden layer l (tanh / sigmoid) in bits 4l .. 4l+3
    int theta_off;
    std::vector<int> sizes;          // n0 .. nL (nL == 1); DGM: {d, modes, 1}
    int kind = 0;                    // 0: Chain of Dense layers; 1: the reference's DGM architecture (src/dgm.jl:97-115)
    int dgm_layers = 0;              // DGM: number of gated (LSTM-type) layers
    int act2 = 0;                    // DGM: activation of the H gate (activation2); `act` is activation1
    // periodic input embedding in front of the chain (Boltz.Layers.PeriodicEmbedding(idxs, periods), reference test
    // test/CUDA/nnpde_cuda__1d_pde_dirichlet_bc_cuda.jl:26,48): inputs emb_idx[k] (0-based) leave the input list and come back at its end
    // as sin(2 pi x / period) for all k, then cos(2 pi x / period) for all k; sizes[0] counts the FEATURES the first Dense layer takes
    std::vector<int> emb_idx;
    std::vector<double> emb_period;
    int n_inputs() const { return sizes[0] - (int)emb_idx.size(); }       // arguments of the dependent variable
    int nparams() const {
        if (kind == 1) { const int d = sizes[0], M = sizes[1]; return M * d + M + dgm_layers * (4 * M * d + 4 * M * M + 4 * M) + M + 1; }
        int n = 0;
        for (size_t i = 0; i + 1 < sizes.size(); ++i) n += sizes[i + 1] * sizes[i] + sizes[i + 1];
        return n;
    }
    int maxhidden() const {
        int m = 0;
        for (size_t i = 1; i + 1 < sizes.size(); ++i) m = std::max(m, sizes[i]);
        return m;
    }
};
struct LinearForm { float k = 0.f; float a[pk::LIN_MAX_C] = {}; float b[pk::LIN_MAX_SRC] = {}; };
struct EmbCol { int src; double omega; int is_cos; };      // device coordinate row = sin / cos (omega x coordinate `src`)
struct Term {
    int d = 0;                       // coordinate rows of the device point set (embedded terms: d_user rows + emb_cols)
    int d_user = 0;                  // coordinates per point at the C ABI (pinn_set_points, samplers, pinn_get_points); == d without embeddings
    std::vector<EmbCol> emb_cols;    // rows d_user .. d-1, written by aux::k_embed whenever the point set changes (descriptor.cpp: apply_embeddings)
    float* d_upts = nullptr;         // embedded terms: the point set as installed, [n][d_user]; otherwise unused (d_pts is the set)
    EmbCol* d_emb_cols = nullptr;
    bool linear = false;             // the fused tape is affine in the jet channels / sources (plan.cpp: detect_linear)
    LinearForm lin;
    std::vector<Slot> slots;
    std::vector<rp::Instr> ops;      // descriptor row numbering
    std::vector<double> imm64;       // the ops' immediates as the descriptor spelled them (Instr::imm is a float): the float64 evaluation mode reads these
    int out_row = 0;
    // plan
    int net = -1;
    int group = -1;
    int slot_in_group = -1;
    int coupled = -1;                // >= 0: index into pinn_engine::coupled (equation couples several networks)
    std::vector<int> chan_of_slot;
    // per referenced network: which of the term's coordinates feed the network's inputs (descriptor `inmap` lines; default
    // identity) — dependent variables of one system may take different arguments (src/discretize.jl:111-131)
    std::map<int, std::vector<int>> inmap;
    // coordinate-only subexpressions hoisted out of the fused tape (analyse_static): evaluated by k_src per point set
    std::vector<rp::Instr> src_prog;     // compact numbering: rows [0,d) coordinates, row d+i = static op i
    std::vector<int> src_root;           // compact row of source j
    std::vector<int> src_of_op;          // per descriptor op: source index, or -1
    std::vector<int> tape_ops;           // descriptor ops that stay in the fused tape, in order
    rp::Instr* d_src_prog = nullptr;
    float* d_src = nullptr;              // [nsrc][n]
    int64_t src_cap = 0;
    // user-supplied per-point data channels (OP_DATA; pinn_set_point_data), valid for the current point set only
    int ndata = 0;
    float* d_data = nullptr;
    int64_t data_n = 0, data_cap = 0;
    // optional quadrature weights of the current point set, stored as sqrt(n_norm * w_i) (pinn_set_point_weights)
    float* d_pw = nullptr;
    int64_t pw_n = 0, pw_cap = 0;
    // data
    float* d_pts = nullptr;
    long long hint_n = 0;            // descriptor `hint`: points the caller expects to install (0: unknown)
    int64_t pts_cap = 0;             // points d_pts has room for (grows, never shrinks: resampled sets of varying size reuse it)
    int64_t n = 0, n_norm = 0;
    float* d_resid = nullptr;
    int64_t resid_cap = 0;
    // on-device sampler: kind 0 = fixed set, 1 = uniform (StochasticTraining), 2 = Latin hypercube (QuasiRandomTraining default),
    // redrawn before every training step
    int sampler = 0;
    float* d_lb = nullptr;
    float* d_ub = nullptr;
    unsigned seed = 0, draws = 0;
};
struct Group {
    int kind = 0;                    // 0: fused (single-network terms); 1: per-network FWD/GRADIN launches of coupled terms;
                                     // 2: TAIL network of a coupled term: forward + residual tape + reverse in one launch (Coupled::tail)
    int net = -1;
    const pk::SpecInfo* spec = nullptr;
    std::vector<int> terms;
    pk::GroupArgs ga;
    rp::Instr* d_prog = nullptr;
    std::vector<int> prog_off, prog_n, out_row;
    float* d_slabs = nullptr;
    double* d_losspart = nullptr;
    float* d_scratch = nullptr;
    size_t scratch_cap = 0;          // family 3: floats allocated (rows x padded points; grows with the point sets)
    float* d_rec = nullptr;          // kind 1, family 2: per-tile records of the forward launch, read back by the reverse launch
    size_t rec_slots = 0;            // (instead of running the forward pass twice; falls back to recomputation above REC_BUDGET)
    bool use_rec = false;
    double* d_tmp = nullptr;         // stage-1 partial sums [nsplit][nent + K]
    std::vector<int> row_theta, row_ptr, row_off;   // host CSR: theta element -> slab offsets of this group
    int* d_ent_theta = nullptr;      // inverse map slab entry -> theta element (-1: none) when every theta element this group feeds has exactly
                                     // one slab entry (families 2, 3): input of the one-kernel reduction (aux::k_reduce_one)
    bool ent_covers_theta = false;   // ... and the group feeds EVERY theta element (single-network problems)
    int nent = 0;
    int slab_floats = 0;             // floats per block of d_slabs
    int blocks = 0;
    int max_blocks = 0;
    bool active = false;
    int chain_to = -1;               // earlier launch group of the same network whose slab set this group may accumulate into
    int merged = -1;                 // >= 0: index into pinn_engine::merged — this group's tiles can ride in ONE launch with another group's
    int launched_blocks = 0;         // workgroups of the launch that carried this group's tiles in the current evaluation
    int launched_by = -1;            // launch group whose launch carried them (itself, or the head of a merged launch)
    plat_event ev_a, ev_b;
    bool timed = false;
};
// an equation that couples several networks (systems of PDEs, src/discretize.jl:58-80): forward launch per network ->
// k_expr (tape over all networks' jets) -> reverse launch per network
struct Coupled {
    int term = -1;
    std::vector<int> nets;           // networks referenced, increasing
    std::vector<int> groups;         // per network: the kind-1 group that runs it
    std::vector<int> slot_net;       // per slot: index into `nets`
    std::vector<float*> d_jets, d_ubar;   // per network: [C_n][N]
    // TAIL launch (r04): the network with the widest channel set of the equation (family 2) runs forward + tape + reverse in ONE launch
    // (kind-2 group, MODE_FUSED): the other networks' jets — written by their forward launches into ONE array, network after network —
    // are its tape's source rows, and it writes their seeds.  No k_expr launch, and the tail network's records never go through HBM.
    // -1: every network takes the forward / k_expr / reverse path (narrow nets, tapes beyond 32 rows, DATA channels, PINN_NO_TAIL_FUSE=1)
    int tail = -1;                   // index into `nets` / `groups`
    std::vector<int> src_off;        // per network: first source row of its kernel's channels (tail: -1)
    int nsrc = 0;                    // source rows = sum of the other kernels' channel counts
    float* d_jets_all = nullptr;     // [nsrc][cap]; d_jets[i] / d_ubar[i] of the other networks point into these (rows packed with stride n)
    float* d_ubar_all = nullptr;
    int64_t cap = 0;
    rp::Instr* d_prog = nullptr;
    double* d_losspart = nullptr;    // pseudo-group for the reduction: [blocks*4][K]
    float* d_pslab = nullptr;        // [blocks][16]
    double* d_tmp = nullptr;
    int blocks = 0, cap_blocks = 0;
    std::vector<int> row_theta, row_ptr, row_off;
};
// two fused launch groups of one network (same shape, slab layout and reduce rows; e.g. interior jet set + value-only boundary set) for
// which a merged kernel exists (pk::PairInfo): one persistent launch walks the head's tiles, then the tail's, with the weight-gradient
// accumulators in registers across both (plan.cpp: plan_merge_groups; engine.cpp: run_loss_grad)
struct MergedUnit {
    int head = -1, tail = -1;
    const pk::PairInfo* pair = nullptr;
    float* d_scratch = nullptr;      // record scratch of the merged launch: max_blocks x pair->SCR
    double* d_losspart = nullptr;    // per-wave loss partials of the merged launch [max_blocks * NW][K] (both groups' terms)
    int max_blocks = 0;
};
struct NetPlan {
    const pk::SpecInfo* spec = nullptr;   // any spec with the right (HP,NHH,D): packed layout is shared
    float* d_packed = nullptr;
    int* d_pack_idx = nullptr;
    const float* cur = nullptr;      // weights the kernels read in the current evaluation: d_packed (DGM kernels: theta + theta_off)
    std::vector<int> h_pack_idx;     // host copy (inverse map construction)
    int npacked = 0;
};

}  // namespace pe

using pe::Coupled; using pe::Group; using pe::MergedUnit; using pe::Net; using pe::NetPlan; using pe::Slot; using pe::Term;

struct pinn_engine {
    int64_t ntheta = 0;
    int np = 0, ne = 0, p_theta_off = 0;
    std::vector<float> p_defaults;
    std::vector<Net> nets;
    std::vector<Term> terms;
    std::vector<Term> terms0;        // the terms as parsed, before the planner's rewrites (Laplacian fusion, source hoisting): what the float64 mode evaluates
    void* f64 = nullptr;             // pe::F64State: the float64 evaluation mode (pinn_set_option "precision"), nullptr = off
    std::vector<Group> groups;
    std::vector<Coupled> coupled;
    std::vector<MergedUnit> merged;
    std::vector<NetPlan> netplans;
    int ncu = 0;
    int gemm = pk::GEMM_SPLIT;       // GEMM arithmetic of the neuron-split kernels of this handle (pinn_set_option "gemm"; $PINN_GEMM at pinn_create)
    bool gemm_auto = false;          // pinn_set_option(h, "gemm", "auto"): the engine picks split / fp32 from the gradient-health figure below
    double grad_health = -1.0;       // rho = |grad sum_k w_k L_k|_2 / sqrt(sum_k w_k L_k) of the last evaluation that measured it (-1: none yet)
    double gemm_delta = -1.0;        // last measured |grad(split) - grad(fp32)|_2 / |grad(fp32)|_2 at one theta (-1: none yet)
    long long gemm_check_t = -1;     // optimiser step / evaluation count of the last such comparison
    long long eval_count = 0;        // host-entry evaluations with a gradient (the "clock" of the policy outside the resident loop)
    int device = 0;                  // the HIP device this handle lives on (pinn_create: the caller's current device; pinn_create_on)
    // engine-owned data-parallel collective (comm.cpp): communicator of this handle's rank, nullptr = single device
    void* comm = nullptr;
    int comm_size = 1, comm_rank = 0;
    bool comm_per_process = false;   // the other ranks live in other processes (pinn_comm_init_rank / _custom): this handle's own calls carry the collective
    pinn_allreduce_fn comm_fn = nullptr;     // caller-supplied transport (pinn_comm_init_custom) instead of RCCL
    void* comm_ctx = nullptr;
    plat_stream stream = nullptr;
    bool own_stream = true;
    float* d_theta = nullptr;
    float* d_params = nullptr;
    float* d_defaults = nullptr;
    double* d_lossraw = nullptr;
    int* d_gr_ptr = nullptr;         // global reduce CSR over theta: contributions (group, entry)
    int* d_gr_grp = nullptr;
    int* d_gr_ent = nullptr;
    int* d_red_perm = nullptr;       // [P + K] thread order of the one-stage reduction (aux::Reduce2Args::perm)
    float* d_out = nullptr;          // [P grad | K raw sums]
    // pinned host block of the host entry points: [P floats theta | P + K floats out | K doubles raw sums]
    float* hp_theta = nullptr;
    float* hp_out = nullptr;
    double* hp_raw = nullptr;
    plat_event ev0, ev1, ev2, ev3;
    plat_stream aux_stream[2] = {nullptr, nullptr};     // under-filled launch groups run concurrently (fork/join by events)
    plat_event ev_fork, ev_join[aux::MAX_GROUPS];
    float last_kernel_ms = 0.f, last_total_ms = 0.f;
    bool timing_valid = false;
    int timing_level = 0;        // 0 (default since r04): no events, 1: per-launch-group kernel events, 2: + phase events (pinn_last_timing)
    int timing_group = -1;       // level >= 1: which launch group gets events (-1: all)
    // resident-theta Adam state
    float* d_opt_theta = nullptr;
    float* d_opt_m = nullptr;
    float* d_opt_v = nullptr;
    float* d_opt_out = nullptr;      // [P + K]
    float* d_w_over_n = nullptr;     // [K]
    double* d_hist = nullptr;
    int hist_cap = 0;
    long long opt_t = 0;
    // device-side step state of the resident loop (pinn_adam_steps): [0] = step index of the current call; draw counters and the
    // sampled-term mask per term; bias-correction table [2 x steps]
    int* d_inv_ptr = nullptr;        // theta element -> positions (net << 24 | offset) in the packed weight images
    int* d_inv_pos = nullptr;
    bool inv_ok = false;
    int* d_step = nullptr;
    unsigned* d_draws = nullptr;
    int* d_sampled = nullptr;
    float* d_c12 = nullptr;
    int c12_cap = 0;
    unsigned* d_bar = nullptr;       // grid-barrier words of the persistent training kernel (pinn_train.hpp)
    unsigned* h_flag = nullptr;      // host-mapped: a launch's barrier timed out
    unsigned bar_arrivals = 0;       // arrivals the barrier counter has seen so far (it is never reset between launches)
    float* d_sums2 = nullptr;        // its [2][K] per-step sums
    int* d_own_r = nullptr;          // its thread -> element map (pinn_train.hpp: TrainArgs::own_r), built for own_blocks workgroups
    int own_blocks = 0, hist_gid = 0;
    void* d_train_samp = nullptr;    // its table of redrawn terms (pk::TrainSampler[train_samp_cap])
    int train_samp_cap = 0;
    float* d_opt_bak = nullptr;      // [3 P] snapshot of (theta, m, v) at the start of a persistent launch (restored when its barrier times out)
    int max_contrib = 0, max_inv_pos = 0;      // most slab entries / image positions of one theta element (plan.cpp)
    bool persistent = true;          // pinn_set_option "persistent": small problems run pinn_adam_steps inside one launch
    int eval_path = 0;               // what the last host-entry loss + gradient evaluation ran: 1 the stand-alone kernels, 2 one launch (eval_fused)
    int adam_path = 0;               // what the last pinn_adam_steps call ran: 0 nothing yet, 1 the stand-alone loop, 2 the persistent kernel
    // phi scratch
    float* d_phi_pts = nullptr;
    float* d_phi_out = nullptr;
    float* d_phi_scr = nullptr;      // family 3: scratch rows of a pinn_phi / pinn_derivative call
    size_t phi_scr_cap = 0;
    int64_t phi_cap = 0;
    int phi_chan = 0;                // jet channels d_phi_out holds per point
};

namespace pe {
// descriptor.cpp
int parse_descriptor(const char* text, pinn_engine& E);
int apply_embeddings(pinn_engine& E);
// sexpr.cpp: the symbolic front end ("pinnir 2"): lhs / rhs of an equation as prefix s-expressions -> jet slots + tape
struct SexprContext {
    std::vector<std::string> params;                         // PDE parameter names, in theta.p order
    std::vector<std::string> depvars;                        // dependent-variable name of net i
    std::vector<std::vector<std::string>> depvar_inputs;     // its argument names (dict_depvar_input)
};
int lower_sexpr_term(const SexprContext& C, const std::vector<std::string>& indvars, const std::string& lhs, const std::string& rhs,
                     int np, Term& T);
// program.cpp
void analyse_static(Term& T, int np);
bool fuse_laplacian(Term& T, int np);
// engine.cpp (shared with comm.cpp)
int run_loss_grad(pinn_engine& E, const float* d_theta, float* d_out, const float* term_w, int only_term, bool timing, double* lossraw = nullptr,
                  bool packed_fresh = false, bool loss_only = false, float* d_sums = nullptr);
int upload_theta(pinn_engine& E, const float* theta, int64_t p);
void sums_from_double(float* d_out_sums, const double* d_raw, int K, plat_stream st);      // (float)raw[k] -> out_sums[k], on the stream
// comm.cpp: sum vec[i] ([P + K] floats) and raw[i] (K doubles) of the ndev handles' ranks over their communicator, in place, each on its
// handle's stream.  ndev == 1: a one-process-per-GPU communicator (RCCL or the caller's transport; no communicator: nothing to do);
// ndev > 1: the handles of one pinn_comm_init_all communicator, one grouped RCCL call
int comm_all_reduce(pinn_engine** es, int ndev, float** vec, double** raw);
int comm_all_reduce_f64(pinn_engine** es, int ndev, double** vec, int64_t count);       // `count` doubles per rank, in place (float64 mode)
// every entry point that touches the device first makes the handle's device current (single-process multi-GPU callers)
struct DeviceScope {
    int prev;
    explicit DeviceScope(int dev) : prev(plat_get_device()) { if (prev != dev) plat_set_device(dev); else prev = -1; }
    ~DeviceScope() { if (prev >= 0) plat_set_device(prev); }
};
// f64.cpp: the float64 evaluation mode (pinn_kernels4.hpp)
int f64_enable(pinn_engine& E);
void f64_destroy(pinn_engine& E);
std::string f64_describe(const pinn_engine& E);
int f64_affine_terms(const pinn_engine& E);      // terms whose residual the matrix-pipe tile kernel evaluates in affine form, without the tape interpreter
int f64_merged(const pinn_engine& E);            // merged launch sequences of the last float64 evaluation (small problems: f64.cpp f64_make_groups)
const char* f64_path(const pinn_engine& E);      // kernels of the last float64 evaluation: "mfma" | "lanes" | "mfma+lanes" | "none" | "off"
int f64_points_changed(pinn_engine& E, int term);
int f64_set_points(pinn_engine& E, int term, const double* pts, int64_t n);
int f64_set_point_data(pinn_engine& E, int term, const double* data);      // nullptr: convert the float rows just installed
int f64_eval(pinn_engine& E, const double* theta, const double* term_w, double* term_losses, double* grad);
int f64_residual(pinn_engine& E, int term, const double* theta, double* r);      // host theta, host r[n_term]
int f64_net_eval(pinn_engine& E, int net, const double* theta, const double* pts, int64_t n, int order, const int* axes, double* out);      // d^order phi_net / dx_axes at host points
int f64_adam_apply(pinn_engine& E, const double* grad_and_sums, double lr, double beta1, double beta2, double eps, const float* term_w, double* loss);
int f64_stencil_enable(pinn_engine& E, bool on);      // pinn_set_option "derivative" = "stencil" | "exact"
bool f64_stencil_on(const pinn_engine& E);
int f64_adam_init(pinn_engine& E, const double* theta);
int f64_adam_get(pinn_engine& E, double* theta);
int f64_adam_steps(pinn_engine& E, int nsteps, double lr, double beta1, double beta2, double eps, const float* term_w, double* loss_history,
                   void (*redraw)(pinn_engine&, pe::Term&));
int f64_points_from_device(pinn_engine& E, int term);
int f64_eval_from_device_f32(pinn_engine& E, const float* d_theta, const float* term_w, float* d_out, bool want_grad);
int f64_eval_from_device_f64(pinn_engine& E, const double* d_theta, const float* term_w, double* d_out);
int f64_eval_sharded_local(pinn_engine& E, const double* theta, const double* term_w, double** d_out);      // host theta -> this device's [gradient | sums] (P + K doubles, device)
int f64_adam_steps_comm(pinn_engine** es, int ndev, int nsteps, double lr, double beta1, double beta2, double eps, const float* term_w, double* loss_history,
                        void (*redraw)(pinn_engine&, pe::Term&));
// plan.cpp
// GEMM arithmetic the kernel look-ups of the calling thread select (family 2 kernels exist as split-operand and fp32 twins): set for the
// duration of an entry point that may look kernels up
int current_gemm();
int current_act_hint();             // activation kind of the network being planned (-1: none): which kernels a hiprtc-specialised member compiles first
struct GemmScope {
    int prev;
    explicit GemmScope(int mode);
    ~GemmScope();
};
int round_hp(int h);
const pk::SpecInfo* find_spec(int HP, int NHH, int D, unsigned need_first, const std::vector<std::pair<int, int>>& need_pairs,
                              unsigned need_hi, std::vector<int>* pair_index, int need_variant = 0, int need_family = 0);
int chan_of(const pk::SpecInfo& s, const Slot& sl);
// find_spec, or — for shapes / jet sets outside the ahead-of-time table — compile, cache and load the kernel (jit.cpp) and look again
const pk::SpecInfo* ensure_spec(const Net& N, unsigned need_first, const std::vector<std::pair<int, int>>& need_pairs, unsigned need_hi,
                                int need_family = 0);
// jit.cpp
int jit_round_hp(int h);
int jit_spec(int HP, int NHH, int D, unsigned D1MASK, unsigned long long PAIRS, int NPAIR, unsigned HI, int variant);
std::string jit_take_launch_error();          // message of a specialised kernel that could not be launched since the last call (empty: none); clears it
// general multi-index jet sets (mixed derivatives of order >= 3, orders 5-6): closed, ordered channel list of a set of requested
// multi-indices (nibble 0 = order, nibbles 1.. = sorted axes) and the kernel generated for it
std::vector<unsigned> gen_close(const std::vector<unsigned>& want);
int jit_spec_gen(int HP, int NHH, int D, const std::vector<unsigned>& channels, int variant);
int jit_spec_dgm(int MP, int L, int D, unsigned D1MASK, unsigned long long PAIRS, int NPAIR, unsigned HI, const std::vector<unsigned>* gen_channels, int act1, int act2);
// plan.cpp: process-wide table of requested general sets; a request travels through the (first, pairs, hi) needs as hi = GEN_FLAG | id
constexpr unsigned GEN_FLAG = 0x80000000u;
int gen_set_id(const std::vector<unsigned>& want);             // id of the (closed) set containing `want`
const std::vector<unsigned>& gen_set(int id);
unsigned slot_mi(const Slot& s);                               // multi-index of a slot
bool slot_is_general(const Slot& s);                           // not representable by the fixed channel categories      // jet channel of a slot in a kernel's channel set (-1: not carried)
// kernel-variant bit a network's activation needs beyond the tanh / sigmoid kernels every spec has (SpecInfo::has_sin)
inline int variant_of(int act) { return act == pk::ACT_SIN ? 1 : (act == pk::ACT_MIXED ? 2 : 0); }
std::string spec_name(const pk::SpecInfo& s);
int build_plan(pinn_engine& E);
void free_plan(pinn_engine& E);            // releases everything build_plan allocated (the terms keep their point sets)
void retile(pinn_engine& E, int gi);
}  // namespace pe
