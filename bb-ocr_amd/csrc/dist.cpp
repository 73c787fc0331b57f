// Multi-GPU entry points of the C ABI (include/bbocr.h): one process per GPU, RCCL over xGMI.  What the Python
// host does through torch.distributed (bb-ocr_amd/dist.py) for a host that has no torch: communicator set-up from a shared unique id,
// ONE broadcast of the packed weight blob, the page scatter as one grouped batch of point-to-point sends (the root's 7 xGMI links carry 7
// different blocks at once) and the gather of the per-rank result bytes.  No collective sits inside the OCR path itself.
// RCCL is reached through dlopen at bbocr_dist_init (librccl.so.1): a single-GPU user of libbbocr.so needs no RCCL on the machine, and a
// process that already carries torch's copy gets that one (same SONAME).
#include "ctx.h"

#include <dlfcn.h>

namespace {
typedef struct { char internal[128]; } rcclUniqueId;          // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* rcclComm_t;
enum { rcclUint8 = 1, rcclInt64 = 4 };                         // ncclDataType_t values (rccl.h)

struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(rcclUniqueId*) = nullptr;
    int (*CommInitRank)(rcclComm_t*, int, rcclUniqueId, int) = nullptr;
    int (*CommDestroy)(rcclComm_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, rcclComm_t, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.h) break;
        }
        if (!r.h) return;
        auto sym = [&](const char* n) { return dlsym(r.h, n); };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.Broadcast = (decltype(r.Broadcast))sym("ncclBroadcast");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
        r.Send = (decltype(r.Send))sym("ncclSend");
        r.Recv = (decltype(r.Recv))sym("ncclRecv");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    });
    return r;
}
bool rccl_ok() {
    const Rccl& r = rccl();
    return r.h && r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.Broadcast && r.AllGather && r.Send && r.Recv && r.GroupStart && r.GroupEnd;
}
void chk(int rc, const char* what) {
    if (rc == 0) return;
    const Rccl& r = rccl();
    fail(BBOCR_ERR_HIP, std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error " + std::to_string(rc)));
}
// contiguous block partition of n units over `world` ranks (bb-ocr_amd/dist.py::shard_range)
void shard_range(long long n, int rank, int world, long long& a, long long& b) {
    const long long q = n / world, r = n % world;
    a = rank * q + std::min<long long>(rank, r);
    b = a + q + (rank < r ? 1 : 0);
}
rcclComm_t comm_of(bbocr_ctx* c) {
    if (!c->dist_comm) fail(BBOCR_ERR_STATE, "bbocr_dist_init has not been called on this context");
    return (rcclComm_t)c->dist_comm;
}
}  // namespace

void dist_release(bbocr_ctx* c) {       // bbocr_destroy
    if (c->dist_comm && rccl_ok()) (void)rccl().CommDestroy((rcclComm_t)c->dist_comm);
    c->dist_comm = nullptr;
}

extern "C" {

int bbocr_dist_unique_id(void* id128) {
    if (!id128) return BBOCR_ERR_ARG;
    if (!rccl_ok()) return BBOCR_ERR_STATE;
    rcclUniqueId id;
    if (rccl().GetUniqueId(&id) != 0) return BBOCR_ERR_HIP;
    memcpy(id128, &id, sizeof(id));
    return BBOCR_OK;
}

int bbocr_dist_init(bbocr_ctx* ctx, int rank, int world, const void* id128) {
    return guarded(ctx, [&](bbocr_ctx* c) {
        if (!id128 || world <= 0 || rank < 0 || rank >= world) fail(BBOCR_ERR_ARG, "bad rank / world / unique id");
        if (c->dist_comm) fail(BBOCR_ERR_STATE, "this context already has a communicator");
        if (!rccl_ok()) fail(BBOCR_ERR_STATE, "librccl.so could not be loaded (multi-GPU entry points need RCCL on the machine)");
        rcclUniqueId id;
        memcpy(&id, id128, sizeof(id));
        rcclComm_t comm = nullptr;
        chk(rccl().CommInitRank(&comm, world, id, rank), "ncclCommInitRank");
        c->dist_comm = comm;
        c->dist_rank = rank;
        c->dist_world = world;
    }, /*exclusive=*/true);
}

int bbocr_dist_finalize(bbocr_ctx* ctx) {
    return guarded(ctx, [&](bbocr_ctx* c) {
        if (!c->dist_comm) return;
        (void)hipStreamSynchronize(c->stream);
        chk(rccl().CommDestroy((rcclComm_t)c->dist_comm), "ncclCommDestroy");
        c->dist_comm = nullptr;
        c->dist_world = 0;
    }, /*exclusive=*/true);
}

// rank `root` holds loaded weights (bbocr_load_weights); every other rank has laid its plans out with bbocr_alloc_weights (same
// precision, same networks).  ONE ncclBroadcast of the packed blob, device to device.
int bbocr_bcast_weights(bbocr_ctx* ctx, int root) {
    return guarded(ctx, [&](bbocr_ctx* c) {
        rcclComm_t comm = comm_of(c);
        if (root < 0 || root >= c->dist_world) fail(BBOCR_ERR_ARG, "bad root");
        if (!c->craft_loaded && !c->crnn_loaded) fail(BBOCR_ERR_STATE, "no weight plans on this rank (bbocr_load_weights on the root, bbocr_alloc_weights elsewhere)");
        // agree on the blob size first: a rank with another precision must fail here, not corrupt its weights
        const size_t bytes = weights_blob_bytes(c);
        DevBuf sizes;
        sizes.ensure(sizeof(long long) * (size_t)(c->dist_world + 1));
        long long mine = (long long)bytes;
        long long* d_all = (long long*)sizes.p;
        HIPCHK(hipMemcpyAsync(d_all + c->dist_world, &mine, sizeof(mine), hipMemcpyHostToDevice, c->stream));
        chk(rccl().AllGather(d_all + c->dist_world, d_all, 1, rcclInt64, comm, c->stream), "ncclAllGather(sizes)");
        std::vector<long long> all((size_t)c->dist_world);
        HIPCHK(hipMemcpyAsync(all.data(), d_all, sizeof(long long) * all.size(), hipMemcpyDeviceToHost, c->stream));
        slot_sync(c, c->stream);
        for (long long v : all)
            if (v != mine) fail(BBOCR_ERR_WEIGHTS, "ranks disagree on the weight blob size: same precision and networks on every rank?");
        DevBuf blob;
        blob.ensure(bytes);
        if (c->dist_rank == root) weights_export(c, blob.p, bytes);
        chk(rccl().Broadcast(blob.p, blob.p, bytes, rcclUint8, root, comm, c->stream), "ncclBroadcast(weights)");
        slot_sync(c, c->stream);
        if (c->dist_rank != root) weights_import(c, blob.p, bytes);
    }, /*exclusive=*/true);
}

// The loader rank `root` holds n_units units of unit_bytes each (pages: H*W*3) contiguously in dev_all; every rank receives its
// contiguous block [first, first + count) into dev_local (capacity: ceil(n_units / world) units).  One grouped batch of sends.
int bbocr_scatter_images(bbocr_ctx* ctx, const uint8_t* dev_all, long long n_units, size_t unit_bytes, int root, uint8_t* dev_local, long long* first,
                         long long* count) {
    return guarded(ctx, [&](bbocr_ctx* c) {
        rcclComm_t comm = comm_of(c);
        if (root < 0 || root >= c->dist_world || n_units < 0 || unit_bytes == 0) fail(BBOCR_ERR_ARG, "bad scatter arguments");
        long long a, b;
        shard_range(n_units, c->dist_rank, c->dist_world, a, b);
        if (first) *first = a;
        if (count) *count = b - a;
        if (b > a && !dev_local) fail(BBOCR_ERR_ARG, "null destination");
        if (c->dist_rank == root) {
            if (n_units > 0 && !dev_all) fail(BBOCR_ERR_ARG, "the root must hold the units");
            chk(rccl().GroupStart(), "ncclGroupStart");
            for (int r = 0; r < c->dist_world; ++r) {
                long long ra, rb;
                shard_range(n_units, r, c->dist_world, ra, rb);
                if (r == root || rb <= ra) continue;
                chk(rccl().Send(dev_all + (size_t)ra * unit_bytes, (size_t)(rb - ra) * unit_bytes, rcclUint8, r, comm, c->stream), "ncclSend");
            }
            chk(rccl().GroupEnd(), "ncclGroupEnd");
            if (b > a) HIPCHK(hipMemcpyAsync(dev_local, dev_all + (size_t)a * unit_bytes, (size_t)(b - a) * unit_bytes, hipMemcpyDeviceToDevice, c->stream));
        } else if (b > a) {
            chk(rccl().GroupStart(), "ncclGroupStart");
            chk(rccl().Recv(dev_local, (size_t)(b - a) * unit_bytes, rcclUint8, root, comm, c->stream), "ncclRecv");
            chk(rccl().GroupEnd(), "ncclGroupEnd");
        }
        slot_sync(c, c->stream);
    }, /*exclusive=*/true);
}

// Every rank contributes `local_bytes` bytes of host memory (its packed results, bbocr_result_pack); on `root`, *all receives a
// malloc'd concatenation in rank order (free with bbocr_free_bytes) and sizes[world] the per-rank byte counts; elsewhere *all = NULL.
int bbocr_gather_results(bbocr_ctx* ctx, const void* local, size_t local_bytes, int root, void** all, size_t* sizes) {
    return guarded(ctx, [&](bbocr_ctx* c) {
        rcclComm_t comm = comm_of(c);
        const int W = c->dist_world, me = c->dist_rank;
        if (root < 0 || root >= W || (local_bytes && !local) || !all) fail(BBOCR_ERR_ARG, "bad gather arguments");
        *all = nullptr;
        DevBuf dsz;
        dsz.ensure(sizeof(long long) * (size_t)(W + 1));
        long long mine = (long long)local_bytes;
        long long* d_all = (long long*)dsz.p;
        HIPCHK(hipMemcpyAsync(d_all + W, &mine, sizeof(mine), hipMemcpyHostToDevice, c->stream));
        chk(rccl().AllGather(d_all + W, d_all, 1, rcclInt64, comm, c->stream), "ncclAllGather(sizes)");
        std::vector<long long> sz((size_t)W);
        HIPCHK(hipMemcpyAsync(sz.data(), d_all, sizeof(long long) * sz.size(), hipMemcpyDeviceToHost, c->stream));
        slot_sync(c, c->stream);
        size_t total = 0;
        std::vector<size_t> off((size_t)W + 1, 0);
        for (int r = 0; r < W; ++r) { off[r + 1] = off[r] + (size_t)sz[r]; if (sizes && me == root) sizes[r] = (size_t)sz[r]; }
        total = off[W];
        DevBuf buf;
        buf.ensure(std::max<size_t>(me == root ? total : local_bytes, 16));
        if (me == root) {
            if (local_bytes) HIPCHK(hipMemcpyAsync((char*)buf.p + off[me], local, local_bytes, hipMemcpyHostToDevice, c->stream));
            chk(rccl().GroupStart(), "ncclGroupStart");
            for (int r = 0; r < W; ++r)
                if (r != root && sz[r] > 0) chk(rccl().Recv((char*)buf.p + off[r], (size_t)sz[r], rcclUint8, r, comm, c->stream), "ncclRecv");
            chk(rccl().GroupEnd(), "ncclGroupEnd");
            void* host = malloc(std::max<size_t>(total, 1));
            if (!host) fail(BBOCR_ERR_INTERNAL, "out of host memory");
            if (total) {
                const hipError_t e = hipMemcpyAsync(host, buf.p, total, hipMemcpyDeviceToHost, c->stream);
                if (e != hipSuccess) { free(host); HIPCHK(e); }
            }
            try { slot_sync(c, c->stream); } catch (...) { free(host); throw; }
            *all = host;
        } else if (local_bytes) {
            HIPCHK(hipMemcpyAsync(buf.p, local, local_bytes, hipMemcpyHostToDevice, c->stream));
            chk(rccl().GroupStart(), "ncclGroupStart");
            chk(rccl().Send(buf.p, local_bytes, rcclUint8, root, comm, c->stream), "ncclSend");
            chk(rccl().GroupEnd(), "ncclGroupEnd");
            slot_sync(c, c->stream);
        }
    }, /*exclusive=*/true);
}

void bbocr_free_bytes(void* p) { free(p); }

// bbocr_result <-> one flat byte block (what bbocr_gather_results moves): [n_images, n_boxes, n_chars] int64, then box_off, quads, is_free,
// text_off, text_idx, conf back to back
int bbocr_result_pack(const bbocr_result* r, void** bytes, size_t* n) {
    if (!r || !bytes || !n) return BBOCR_ERR_ARG;
    const long long B = r->n_images, nb = r->box_off[B], nt = r->text_off[nb];
    const size_t total = 3 * sizeof(long long) + (size_t)(B + 1) * 4 + (size_t)nb * 64 + (size_t)nb * 4 + (size_t)(nb + 1) * 4 + (size_t)nt * 4 + (size_t)nb * 8;
    char* p = (char*)malloc(total);
    if (!p) return BBOCR_ERR_INTERNAL;
    char* q = p;
    auto put = [&](const void* src, size_t k) { memcpy(q, src, k); q += k; };
    const long long hdr[3] = {B, nb, nt};
    put(hdr, sizeof(hdr));
    put(r->box_off, (size_t)(B + 1) * 4);
    put(r->quads, (size_t)nb * 64);
    put(r->is_free, (size_t)nb * 4);
    put(r->text_off, (size_t)(nb + 1) * 4);
    put(r->text_idx, (size_t)nt * 4);
    put(r->conf, (size_t)nb * 8);
    *bytes = p;
    *n = total;
    return BBOCR_OK;
}

int bbocr_result_unpack(const void* bytes, size_t n, bbocr_result** out) {
    if (!bytes || !out || n < 3 * sizeof(long long)) return BBOCR_ERR_ARG;
    const char* q = (const char*)bytes;
    long long hdr[3];
    memcpy(hdr, q, sizeof(hdr));
    q += sizeof(hdr);
    const long long B = hdr[0], nb = hdr[1], nt = hdr[2];
    if (B < 0 || nb < 0 || nt < 0) return BBOCR_ERR_ARG;
    const size_t total = 3 * sizeof(long long) + (size_t)(B + 1) * 4 + (size_t)nb * 64 + (size_t)nb * 4 + (size_t)(nb + 1) * 4 + (size_t)nt * 4 + (size_t)nb * 8;
    if (total != n) return BBOCR_ERR_ARG;
    bbocr_result* r = (bbocr_result*)calloc(1, sizeof(bbocr_result));
    if (!r) return BBOCR_ERR_INTERNAL;
    r->n_images = (int)B;
    auto take = [&](size_t k) { void* d = malloc(std::max<size_t>(k, 1)); if (d) memcpy(d, q, k); q += k; return d; };
    r->box_off = (int*)take((size_t)(B + 1) * 4);
    r->quads = (double*)take((size_t)nb * 64);
    r->is_free = (int*)take((size_t)nb * 4);
    r->text_off = (int*)take((size_t)(nb + 1) * 4);
    r->text_idx = (int*)take((size_t)nt * 4);
    r->conf = (double*)take((size_t)nb * 8);
    if (!r->box_off || !r->quads || !r->is_free || !r->text_off || !r->text_idx || !r->conf) { bbocr_free_result(r); return BBOCR_ERR_INTERNAL; }
    if (r->box_off[B] != nb || r->text_off[nb] != nt) { bbocr_free_result(r); return BBOCR_ERR_ARG; }      // offsets must match the header
    *out = r;
    return BBOCR_OK;
}

}  // extern "C"
