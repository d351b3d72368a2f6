// comm.cpp — the engine-owned data-parallel collective: every term's point set is sharded over the GPUs of one node,
// theta is replicated, and ONE all-reduce (sum, fp32) of the packed vector [gradient (P) | per-term sums of squares (K)] completes an
// evaluation.  The reference has no collective: it aggregates the K per-term losses on the host (src/discretize.jl:568-588) — this is
// the multi-GPU form of that aggregation, issued by the engine on the evaluation's own stream (RCCL over xGMI; the message is
// 51 KB - 0.8 MB, i.e. latency-bound, so it is ONE collective per evaluation and never split into buckets).
// The K sums of squares additionally travel as DOUBLES (K x 8 bytes, grouped into the same RCCL launch): every rank's sums are exact
// doubles (fixed-order device reduction), so the N-rank losses equal the single-device ones to double rounding.
//
// RCCL is bound lazily (dlopen of librccl.so.1 on first use): libpinn_hip.so loads and runs single-GPU work without it, and in a
// process that already carries an RCCL (e.g. PyTorch's bundled copy) the same instance is shared instead of a second one being mapped.
//
// Two ways to form the communicator:
//   * one process per GPU (torchrun / MPI style): rank 0 calls pinn_comm_unique_id, the host side distributes the 128 bytes by any means
//     (file, TCP store, MPI_Bcast), every rank calls pinn_comm_init_rank, then pinn_loss_grad_sharded_device per evaluation;
//   * one process, several devices (what a Julia caller does): pinn_create_on(desc, device_i) for every device, pinn_comm_init_all
//     over the handles (ncclCommInitAll), then pinn_loss_grad_sharded(handles, ...) per evaluation (host theta in, loss + gradient out).
#include "engine_types.hpp"

using namespace pe;

#ifndef PINN_EMU
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};
Rccl& rccl() {
    static Rccl R = [] {
        Rccl r;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (!r.lib) { r.err = std::string("RCCL is not loadable (") + dlerror() + ")"; return r; }
        auto sym = [&](const char* n) { void* p = dlsym(r.lib, n); if (!p && r.err.empty()) r.err = std::string("RCCL lacks symbol ") + n; return p; };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        return r;
    }();
    return R;
}
int nccl_fail(const char* what, ncclResult_t rc) { return fail(std::string(what) + ": " + rccl().GetErrorString(rc)); }
}  // namespace
#define NCCL_TRY(call, what) do { ncclResult_t rc_ = (call); if (rc_ != ncclSuccess) return nccl_fail(what, rc_); } while (0)
#else
// Emulation build (tests only): "devices" are labels on host memory, so the single-process form (pinn_comm_init_all +
// pinn_loss_grad_sharded) is emulated by summing the members' output vectors in rank order; the one-process-per-GPU form needs a
// real transport and is refused unless the communicator has a single rank.
namespace {
struct EmuComm { std::vector<pinn_engine*> members; };
}
#endif

static int check_shards(pinn_engine& E) {
    for (size_t t = 0; t < E.terms.size(); ++t)
        if (E.terms[t].n_norm < E.terms[t].n) return fail("term " + std::to_string(t) + ": n_norm (the global point count) is smaller than the shard");
    return 0;
}

extern "C" {

int pinn_comm_unique_id(void* id, int64_t nbytes) {
    if (!id || nbytes < PINN_COMM_ID_BYTES) return fail("pinn_comm_unique_id: the id buffer must hold PINN_COMM_ID_BYTES (128) bytes");
#ifndef PINN_EMU
    static_assert(sizeof(ncclUniqueId) == PINN_COMM_ID_BYTES, "ncclUniqueId size");
    if (!rccl().err.empty()) return fail(rccl().err);
    ncclUniqueId u;
    NCCL_TRY(rccl().GetUniqueId(&u), "ncclGetUniqueId");
    std::memcpy(id, &u, sizeof u);
#else
    std::memset(id, 0x5a, PINN_COMM_ID_BYTES);
#endif
    return 0;
}

int pinn_comm_init_rank(pinn_handle h, int nranks, int rank, const void* id, int64_t nbytes) {
    if (!h || !id || nbytes < PINN_COMM_ID_BYTES) return fail("pinn_comm_init_rank: null handle / id buffer too small");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail("pinn_comm_init_rank: rank out of range");
    pinn_engine& E = *h;
    if (E.comm) return fail("pinn_comm_init_rank: the handle already belongs to a communicator (pinn_comm_destroy first)");
    DeviceScope scope(E.device);
#ifndef PINN_EMU
    if (!rccl().err.empty()) return fail(rccl().err);
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    ncclComm_t c = nullptr;
    NCCL_TRY(rccl().CommInitRank(&c, nranks, u, rank), "ncclCommInitRank");
    E.comm = c;
#else
    if (nranks != 1) return fail("pinn_comm_init_rank: the emulation build has no inter-process transport (nranks must be 1)");
    EmuComm* c = new EmuComm();
    c->members.push_back(&E);
    E.comm = c;
#endif
    E.comm_size = nranks;
    E.comm_rank = rank;
    E.comm_per_process = true;
    return 0;
}

// bring-your-own-collective communicator (MPI, torch.distributed, ...): the engine calls `fn` where it would call ncclAllReduce
namespace { int custom_tag; }
int pinn_comm_init_custom(pinn_handle h, int nranks, int rank, pinn_allreduce_fn fn, void* ctx) {
    if (!h || !fn) return fail("pinn_comm_init_custom: null handle / callback");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail("pinn_comm_init_custom: rank out of range");
    pinn_engine& E = *h;
    if (E.comm) return fail("pinn_comm_init_custom: the handle already belongs to a communicator (pinn_comm_destroy first)");
    E.comm = &custom_tag;
    E.comm_fn = fn;
    E.comm_ctx = ctx;
    E.comm_size = nranks;
    E.comm_rank = rank;
    E.comm_per_process = true;
    return 0;
}

int pinn_comm_init_all(pinn_handle* hs, int ndev) {
    if (!hs || ndev < 1) return fail("pinn_comm_init_all: need at least one handle");
    for (int i = 0; i < ndev; ++i) {
        if (!hs[i]) return fail("pinn_comm_init_all: null handle");
        if (hs[i]->comm) return fail("pinn_comm_init_all: handle " + std::to_string(i) + " already belongs to a communicator");
        if (hs[i]->ntheta != hs[0]->ntheta || hs[i]->terms.size() != hs[0]->terms.size())
            return fail("pinn_comm_init_all: the handles were not created from the same descriptor");
        for (int j = 0; j < i; ++j)
            if (hs[j]->device == hs[i]->device && ndev > 1
#ifdef PINN_EMU
                && false
#endif
            ) return fail("pinn_comm_init_all: handles " + std::to_string(j) + " and " + std::to_string(i) + " live on the same device");
    }
#ifndef PINN_EMU
    if (!rccl().err.empty()) return fail(rccl().err);
    std::vector<int> devs(ndev);
    std::vector<ncclComm_t> comms(ndev, nullptr);
    for (int i = 0; i < ndev; ++i) devs[i] = hs[i]->device;
    NCCL_TRY(rccl().CommInitAll(comms.data(), ndev, devs.data()), "ncclCommInitAll");
    for (int i = 0; i < ndev; ++i) { hs[i]->comm = comms[i]; hs[i]->comm_size = ndev; hs[i]->comm_rank = i; }
#else
    EmuComm* c = new EmuComm();
    for (int i = 0; i < ndev; ++i) c->members.push_back(hs[i]);
    for (int i = 0; i < ndev; ++i) { hs[i]->comm = c; hs[i]->comm_size = ndev; hs[i]->comm_rank = i; }
#endif
    return 0;
}

int pinn_comm_size(pinn_handle h) { return h ? h->comm_size : -1; }
int pinn_comm_rank(pinn_handle h) { return h ? h->comm_rank : -1; }

int pinn_comm_destroy(pinn_handle h) {
    if (!h || !h->comm) return 0;
    if (h->comm_fn) {
        DeviceScope scope(h->device);
        plat_sync(h->stream);
        h->comm = nullptr; h->comm_fn = nullptr; h->comm_ctx = nullptr;
        h->comm_size = 1; h->comm_rank = 0; h->comm_per_process = false;
        return 0;
    }
#ifndef PINN_EMU
    DeviceScope scope(h->device);
    plat_sync(h->stream);
    (void)rccl().CommDestroy((ncclComm_t)h->comm);
#else
    EmuComm* c = (EmuComm*)h->comm;
    for (auto it = c->members.begin(); it != c->members.end(); ++it)
        if (*it == h) { c->members.erase(it); break; }
    if (c->members.empty()) delete c;
#endif
    h->comm = nullptr;
    h->comm_size = 1;
    h->comm_rank = 0;
    h->comm_per_process = false;
    return 0;
}

}  // extern "C"

int pe::comm_all_reduce(pinn_engine** es, int ndev, float** vec, double** raw) {
    const int64_t P = es[0]->ntheta;
    const int K = (int)es[0]->terms.size();
    if (ndev == 1) {
        pinn_engine& E = *es[0];
        if (!E.comm) return 0;
        if (E.comm_fn) {
            if (E.comm_fn(E.comm_ctx, vec[0], P + K, 0, (void*)E.stream)) return fail("the caller's all-reduce callback failed (gradient vector)");
            if (E.comm_fn(E.comm_ctx, raw[0], K, 1, (void*)E.stream)) return fail("the caller's all-reduce callback failed (loss sums)");
            return 0;
        }
#ifndef PINN_EMU
        NCCL_TRY(rccl().GroupStart(), "ncclGroupStart");
        ncclResult_t r1 = rccl().AllReduce(vec[0], vec[0], (size_t)(P + K), ncclFloat, ncclSum, (ncclComm_t)E.comm, E.stream);
        ncclResult_t r2 = rccl().AllReduce(raw[0], raw[0], (size_t)K, ncclDouble, ncclSum, (ncclComm_t)E.comm, E.stream);
        NCCL_TRY(rccl().GroupEnd(), "ncclGroupEnd");
        if (r1 != ncclSuccess) return nccl_fail("ncclAllReduce", r1);
        if (r2 != ncclSuccess) return nccl_fail("ncclAllReduce", r2);
#else
        if (E.comm_size != 1) return fail("the emulation build has no inter-process transport of its own (pinn_comm_init_custom supplies one)");
#endif
        return 0;
    }
#ifndef PINN_EMU
    NCCL_TRY(rccl().GroupStart(), "ncclGroupStart");
    for (int i = 0; i < ndev; ++i) {
        pinn_engine& E = *es[i];
        DeviceScope scope(E.device);
        ncclResult_t rc = rccl().AllReduce(vec[i], vec[i], (size_t)(P + K), ncclFloat, ncclSum, (ncclComm_t)E.comm, E.stream);
        if (rc == ncclSuccess) rc = rccl().AllReduce(raw[i], raw[i], (size_t)K, ncclDouble, ncclSum, (ncclComm_t)E.comm, E.stream);
        if (rc != ncclSuccess) { (void)rccl().GroupEnd(); return nccl_fail("ncclAllReduce", rc); }
    }
    NCCL_TRY(rccl().GroupEnd(), "ncclGroupEnd");
#else
    {
        std::vector<float> sum((size_t)(P + K), 0.f);
        std::vector<double> rs((size_t)K, 0.0);
        for (int i = 0; i < ndev; ++i) {
            for (int64_t j = 0; j < P + K; ++j) sum[j] += vec[i][j];          // rank order: deterministic
            for (int k = 0; k < K; ++k) rs[k] += raw[i][k];
        }
        for (int i = 0; i < ndev; ++i) {
            std::memcpy(vec[i], sum.data(), sizeof(float) * (P + K));
            std::memcpy(raw[i], rs.data(), sizeof(double) * K);
        }
    }
#endif
    return 0;
}

// the same for `count` DOUBLES per rank (the float64 evaluation mode: [gradient | sums] in double, r06)
int pe::comm_all_reduce_f64(pinn_engine** es, int ndev, double** vec, int64_t count) {
    if (ndev == 1) {
        pinn_engine& E = *es[0];
        if (!E.comm) return 0;
        if (E.comm_fn) {
            if (E.comm_fn(E.comm_ctx, vec[0], count, 1, (void*)E.stream)) return fail("the caller's all-reduce callback failed (float64 gradient vector)");
            return 0;
        }
#ifndef PINN_EMU
        const ncclResult_t r1 = rccl().AllReduce(vec[0], vec[0], (size_t)count, ncclDouble, ncclSum, (ncclComm_t)E.comm, E.stream);
        if (r1 != ncclSuccess) return nccl_fail("ncclAllReduce", r1);
#else
        if (E.comm_size != 1) return fail("the emulation build has no inter-process transport of its own (pinn_comm_init_custom supplies one)");
#endif
        return 0;
    }
#ifndef PINN_EMU
    NCCL_TRY(rccl().GroupStart(), "ncclGroupStart");
    for (int i = 0; i < ndev; ++i) {
        pinn_engine& E = *es[i];
        DeviceScope scope(E.device);
        const ncclResult_t rc = rccl().AllReduce(vec[i], vec[i], (size_t)count, ncclDouble, ncclSum, (ncclComm_t)E.comm, E.stream);
        if (rc != ncclSuccess) { (void)rccl().GroupEnd(); return nccl_fail("ncclAllReduce", rc); }
    }
    NCCL_TRY(rccl().GroupEnd(), "ncclGroupEnd");
#else
    {
        std::vector<double> sum((size_t)count, 0.0);
        for (int i = 0; i < ndev; ++i)
            for (int64_t j = 0; j < count; ++j) sum[(size_t)j] += vec[i][j];          // rank order: deterministic
        for (int i = 0; i < ndev; ++i) std::memcpy(vec[i], sum.data(), sizeof(double) * (size_t)count);
    }
#endif
    return 0;
}

extern "C" {

// float64 mode over a communicator (r06): this rank's shards evaluated by the double kernels, then ONE all-reduce of [P + K] doubles in place
int pinn_loss_grad_sharded_device_f64(pinn_handle h, const double* d_theta, const float* term_w, double* d_out, void* stream) {
    if (!h || !d_theta || !d_out) return fail("pinn_loss_grad_sharded_device_f64: null argument");
    pinn_engine& E = *h;
    if (!E.f64) return fail("pinn_loss_grad_sharded_device_f64: the handle is not in the float64 evaluation mode (pinn_set_option(h, \"precision\", \"f64\"))");
    if (!E.comm) return fail("pinn_loss_grad_sharded_device_f64: the handle has no communicator (pinn_comm_init_rank / pinn_comm_init_custom)");
    if (!E.comm_per_process && E.comm_size > 1)
        return fail("pinn_loss_grad_sharded_device_f64: the handle belongs to a single-process communicator (pinn_comm_init_all): use pinn_loss_grad_sharded_f64");
    DeviceScope scope(E.device);
    if (check_shards(E)) return 1;
    plat_stream saved = E.stream;
    E.stream = (plat_stream)stream;
    int rc = f64_eval_from_device_f64(E, d_theta, term_w, d_out);
    if (!rc) {
        pinn_engine* es[1] = {&E};
        double* vec[1] = {d_out};
        rc = comm_all_reduce_f64(es, 1, vec, E.ntheta + (int64_t)E.terms.size());
    }
    E.stream = saved;
    return rc;
}

int pinn_loss_grad_sharded_f64(pinn_handle* hs, int ndev, const double* theta, int64_t p, const double* term_w, double* term_losses, double* grad) {
    if (!hs || ndev < 1 || !theta) return fail("pinn_loss_grad_sharded_f64: null argument");
    for (int i = 0; i < ndev; ++i) {
        if (!hs[i] || !hs[i]->comm || hs[i]->comm_size != ndev || hs[i]->comm_rank != i)
            return fail("pinn_loss_grad_sharded_f64: pass the handles of one pinn_comm_init_all communicator, in rank order");
        if (!hs[i]->f64) return fail("pinn_loss_grad_sharded_f64: every handle must be in the float64 evaluation mode");
        if (p != hs[i]->ntheta) return fail("pinn_loss_grad_sharded_f64: theta length mismatch");
    }
    const int64_t P = hs[0]->ntheta;
    const int K = (int)hs[0]->terms.size();
    std::vector<double*> vec(ndev);
    for (int i = 0; i < ndev; ++i) {
        pinn_engine& E = *hs[i];
        DeviceScope scope(E.device);
        if (check_shards(E)) return 1;
        if (f64_eval_sharded_local(E, theta, term_w, &vec[i])) return 1;          // [gradient | sums] of this device's shards, left on the device
    }
    if (comm_all_reduce_f64(hs, ndev, vec.data(), P + K)) return 1;
    std::vector<double> out((size_t)(P + K));
    {
        pinn_engine& E0 = *hs[0];
        DeviceScope scope(E0.device);
        if (plat_d2h(out.data(), vec[0], sizeof(double) * (size_t)(P + K), E0.stream)) return fail("D2H copy failed");
    }
    for (int i = 0; i < ndev; ++i) {
        DeviceScope scope(hs[i]->device);
        if (plat_sync(hs[i]->stream)) return fail(std::string("device error: ") + plat_last_error());
    }
    if (term_losses) for (int k = 0; k < K; ++k) term_losses[k] = out[(size_t)P + k] / (double)hs[0]->terms[k].n_norm;
    if (grad) std::memcpy(grad, out.data(), sizeof(double) * (size_t)P);
    return 0;
}

int pinn_loss_grad_sharded_device(pinn_handle h, const float* d_theta, const float* term_w, float* d_out, void* stream) {
    if (!h || !d_theta || !d_out) return fail("pinn_loss_grad_sharded_device: null argument");
    pinn_engine& E = *h;
    if (!E.comm) return fail("pinn_loss_grad_sharded_device: the handle has no communicator (pinn_comm_init_rank / pinn_comm_init_all)");
    DeviceScope scope(E.device);
    if (check_shards(E)) return 1;
    plat_stream st = (plat_stream)stream;
    plat_stream saved = E.stream;
    E.stream = st;
    int rc = run_loss_grad(E, d_theta, d_out, term_w, -1, true);
    E.stream = saved;
    if (rc) return rc;
    E.timing_valid = E.timing_level >= 2;
    if (!E.comm_per_process && E.comm_size > 1)
        return fail("pinn_loss_grad_sharded_device: the handle belongs to a single-process communicator (pinn_comm_init_all): use pinn_loss_grad_sharded");
    {
        pinn_engine* es[1] = {&E};
        float* vec[1] = {d_out};
        double* raw[1] = {E.d_lossraw};
        E.stream = st;
        rc = comm_all_reduce(es, 1, vec, raw);
        E.stream = saved;
        if (rc) return rc;
    }
    sums_from_double(d_out + E.ntheta, E.d_lossraw, (int)E.terms.size(), st);      // the exact (double) sums replace the float-summed ones
    return 0;
}

int pinn_loss_grad_sharded(pinn_handle* hs, int ndev, const float* theta, int64_t p, const float* term_w, double* term_losses, float* grad) {
    if (!hs || ndev < 1 || !theta) return fail("pinn_loss_grad_sharded: null argument");
    for (int i = 0; i < ndev; ++i) {
        if (!hs[i] || !hs[i]->comm || hs[i]->comm_size != ndev || hs[i]->comm_rank != i)
            return fail("pinn_loss_grad_sharded: pass the handles of one pinn_comm_init_all communicator, in rank order");
        if (i > 0 && hs[i]->comm != hs[0]->comm
#ifndef PINN_EMU
            && false          // RCCL: one ncclComm_t per rank
#endif
        ) return fail("pinn_loss_grad_sharded: handles of different communicators");
    }
    const int64_t P = hs[0]->ntheta;
    const int K = (int)hs[0]->terms.size();
    // every device: upload theta, evaluate its shards (all asynchronous on the handle's own stream)
    for (int i = 0; i < ndev; ++i) {
        pinn_engine& E = *hs[i];
        DeviceScope scope(E.device);
        if (check_shards(E) || upload_theta(E, theta, p)) return 1;
        if (run_loss_grad(E, E.d_theta, E.d_out, term_w, -1, false)) return 1;
    }
    // one grouped all-reduce of [gradient | sums] across the devices, each rank's call on that rank's stream
    {
        std::vector<float*> vec(ndev);
        std::vector<double*> raw(ndev);
        for (int i = 0; i < ndev; ++i) { vec[i] = hs[i]->d_out; raw[i] = hs[i]->d_lossraw; }
        if (comm_all_reduce(hs, ndev, vec.data(), raw.data())) return 1;
    }
    // the result is identical on every device; rank 0 delivers it
    pinn_engine& E0 = *hs[0];
    {
        DeviceScope scope(E0.device);
        if (plat_d2h(E0.hp_out, E0.d_out, sizeof(float) * (P + K), E0.stream)) return fail("D2H copy failed");
        if (plat_d2h(E0.hp_raw, E0.d_lossraw, sizeof(double) * K, E0.stream)) return fail("D2H copy failed");
    }
    for (int i = 0; i < ndev; ++i) {
        DeviceScope scope(hs[i]->device);
        if (plat_sync(hs[i]->stream)) return fail(std::string("device error: ") + plat_last_error());
    }
    if (term_losses)
        for (int k = 0; k < K; ++k) term_losses[k] = E0.hp_raw[k] / (double)E0.terms[k].n_norm;      // exact double sums, as on one device
    if (grad) std::memcpy(grad, E0.hp_out, sizeof(float) * P);
    return 0;
}

}  // extern "C"
