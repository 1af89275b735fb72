// ff_dist.cpp — multi-GPU frames behind the C ABI (include/firefly/ff_api.h, "multi-GPU" section).
//
// The reference renders on one GPU (its only trace of more is a dead `const bool multi_gpu`, utilities.h:484-487).  Pixels
// are independent, so a frame shards with no exchange while it renders: the image is cut into strips of a few rows, dealt
// round-robin to the GPUs; the scene is replicated; the random numbers are keyed on the global pixel index, so the strips
// are the rows of the one-GPU frame bit for bit.  The only exchange is the gather of the finished strips on the GPU that
// owns the display buffer (the GL pixel buffer of kernel.cu:335-351).  Two shapes of the same thing:
//
//   * one process per GPU (ff_dist_init / ff_render_distributed): an RCCL communicator per state; per frame each rank
//     renders its strips into ONE packed buffer (float3 radiance rows, then rgb8 rows: 15 bytes per pixel) and sends it to
//     rank 0 with a single ncclSend; rank 0 posts one ncclRecv per peer in the same group, straight into the gather
//     buffer (its own strips are rendered in place there), and one kernel scatters all strips to image order.  xGMI is
//     point-to-point: the seven peers' sends travel over seven different links at once, there is no ring to pace them.
//   * one process, several GPUs (ff_multi_*): what a single-process viewer (the reference's main(), kernel.cu:223-368) can
//     call.  Same packing and scatter; the transport is ncclCommInitAll + grouped send/recv, or hipMemcpyPeerAsync when
//     a device appears more than once in the list (RCCL refuses that; it is how a one-GPU box rehearses the path).
//
// RCCL is loaded with dlopen when the first communicator is made: single-GPU users of the library do not need it.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include <dlfcn.h>

#include <hip/hip_runtime.h>
#include <hip/hip_gl_interop.h>
#include <rccl/rccl.h>

#include "ff_state.h"

using namespace ff;

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    // optional (absent from very old libraries: the waits then have a deadline but no early error report / abort)
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
};

Rccl g_rccl;

int load_rccl()
{
    if (g_rccl.handle) return FF_OK;
    const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    void* h = nullptr;
    for (const char* n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return fail(FF_ERR_COMM, "RCCL is not available: %s", dlerror());
    Rccl r;
    r.handle = h;
#define FF_SYM(field, name)                                                                   \
    *reinterpret_cast<void**>(&r.field) = dlsym(h, name);                                     \
    if (!r.field) return fail(FF_ERR_COMM, "RCCL library lacks %s", name);
    FF_SYM(GetUniqueId, "ncclGetUniqueId")
    FF_SYM(CommInitRank, "ncclCommInitRank")
    FF_SYM(CommInitAll, "ncclCommInitAll")
    FF_SYM(CommDestroy, "ncclCommDestroy")
    FF_SYM(GroupStart, "ncclGroupStart")
    FF_SYM(GroupEnd, "ncclGroupEnd")
    FF_SYM(Send, "ncclSend")
    FF_SYM(Recv, "ncclRecv")
    FF_SYM(AllReduce, "ncclAllReduce")
    FF_SYM(GetErrorString, "ncclGetErrorString")
#undef FF_SYM
    *reinterpret_cast<void**>(&r.CommGetAsyncError) = dlsym(h, "ncclCommGetAsyncError");
    *reinterpret_cast<void**>(&r.CommAbort) = dlsym(h, "ncclCommAbort");
    g_rccl = r;
    return FF_OK;
}

#define FF_NCCL(call)                                                                                                   \
    do {                                                                                                                \
        ncclResult_t _r = (call);                                                                                       \
        if (_r != ncclSuccess) return fail(FF_ERR_COMM, "%s failed: %s (%s:%d)", #call, g_rccl.GetErrorString(_r), __FILE__, __LINE__); \
    } while (0)

size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

// Packed strips of one part: [radiance: rows x W float3][rgb8: rows x W x 3 bytes], each section padded to 16 bytes.
struct PackLayout {
    int width = 0, height = 0, strip_rows = 1, num_parts = 1;
    size_t rad_bytes(int part) const { return align16((size_t)ff_strips_local_rows(height, strip_rows, part, num_parts) * (size_t)width * 12); }
    size_t rgb_bytes(int part) const { return align16((size_t)ff_strips_local_rows(height, strip_rows, part, num_parts) * (size_t)width * 3); }
    size_t part_bytes(int part) const { return rad_bytes(part) + rgb_bytes(part); }
    size_t part_offset(int part) const
    {
        size_t off = 0;
        for (int q = 0; q < part; ++q) off += part_bytes(q);
        return off;
    }
    size_t total_bytes() const { return part_offset(num_parts); }
};

// Strip height for `world` parts of a frame of `height` rows: the one - of 1 .. 16 rows - whose LARGEST part has the fewest rows, the
// thinnest such of at least two rows if there is one.  1080 rows: 2-row strips for 2 or 4 ranks, 3-row strips for 8 (135 rows each); with the
// 16 / 8 / 4-row strips of rounds 2-3 six of eight ranks had 136 rows and two had 132, and the slowest rank sets the frame time
// (same box, tools/strip_scaling.py: 98.9 / 97.4 / 96.4 -> 99.5 / 99.1 / 97.5 % of ideal with the 8-sample tail items that went in
// with it, profiles/r04_p_*).  Thin strips also deal neighbouring rows - similar cost - to different ranks.
int default_strip_rows(int world, int height)
{
    if (world <= 1 || height <= 0) return 16;
    int best = 16;
    long best_rows = -1;
    for (int s = 16; s >= 1; --s) {
        long worst = 0;
        for (int p = 0; p < world; ++p) worst = std::max<long>(worst, ff_strips_local_rows(height, s, p, world));
        // (descending: on equal rows the thinner strip replaces the thicker one, except that a 1-row strip only wins outright)
        if (best_rows < 0 || worst < best_rows || (worst == best_rows && s >= 2)) {
            best_rows = worst;
            best = s;
        }
    }
    return best;
}

// Device buffers for the frame's final outputs on the gathering device when the caller passed host pointers (or null).
struct RootOutputs {
    unsigned char* rgb8 = nullptr;
    float* radiance = nullptr;
};

int root_outputs(FfState* s, const FfRenderParams* prm, void* rgb8, int rgb8_on_device, float* radiance, int radiance_on_device, RootOutputs& out)
{
    const size_t pixels = (size_t)prm->width * (size_t)prm->height;
    if (rgb8) {
        if (rgb8_on_device) out.rgb8 = (unsigned char*)rgb8;
        else {
            const int st = ensure_bytes((void**)&s->d_rgb8, &s->rgb8_bytes, pixels * 3 + 16);
            if (st != FF_OK) return st;
            out.rgb8 = s->d_rgb8;
        }
    }
    if (radiance) {
        if (radiance_on_device) out.radiance = radiance;
        else {
            const int st = ensure_bytes((void**)&s->d_radiance, &s->radiance_bytes, pixels * 12 + 16);
            if (st != FF_OK) return st;
            out.radiance = s->d_radiance;
        }
    }
    return FF_OK;
}

int copy_root_outputs_to_host(FfState* s, const FfRenderParams* prm, void* rgb8, int rgb8_on_device, float* radiance, int radiance_on_device,
                              const RootOutputs& out)
{
    const size_t pixels = (size_t)prm->width * (size_t)prm->height;
    if (rgb8 && !rgb8_on_device) FF_HIP(hipMemcpy(rgb8, out.rgb8, pixels * 3, hipMemcpyDeviceToHost));
    if (radiance && !radiance_on_device) FF_HIP(hipMemcpy(radiance, out.radiance, pixels * 12, hipMemcpyDeviceToHost));
    (void)s;
    return FF_OK;
}

} // namespace

// One rank of a process-per-GPU job.
struct FfDistContext {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    bool self_loop = false; // FF_DIST_SELF_LOOP=1: rank 0 also moves its OWN strips through ncclSend/ncclRecv (a one-rank
                            // communicator then exercises the whole transport on a one-GPU box)
    bool broken = false;    // the transport failed or timed out: the communicator was aborted, ff_dist_init makes a new one
    double timeout_s = 300.0; // FF_DIST_TIMEOUT_S: longest wait for the OTHER ranks in one frame (the clock starts when this rank's own
                              // strips have drained: a long frame is not a missing peer)
    int fail_rank = -1;       // FF_DEBUG_DIST_FAIL_RANK (tests, read once at ff_dist_init): that rank reports an injected local failure
    hipEvent_t ev_local = nullptr; // recorded behind this rank's own launches of a frame
    int* d_status = nullptr;  // [0] this rank's status of the frame, [1] the job's (all-reduce, maximum)
    int* h_status = nullptr;  // pinned mirror
    unsigned char* d_pack = nullptr; // this rank's packed strips (non-root ranks; rank 0 with self_loop)
    size_t pack_bytes = 0;
    unsigned char* d_gather = nullptr; // rank 0: every part's packed strips, part after part
    size_t gather_bytes = 0;
    double last_gather_ms = 0.0;
};

namespace {

// The transport is beyond repair (a peer died, a wait ran out): end the communicator's pending work so that this process can
// report the error and leave instead of sitting in a stream wait for ever.  A fresh process is the retry.
void abandon_comm(FfDistContext* d)
{
    // (without ncclCommAbort the handle is dropped, not destroyed: ncclCommDestroy waits for the pending operations, which is
    // the wait this function exists to end)
    if (d->comm && g_rccl.CommAbort) (void)g_rccl.CommAbort(d->comm);
    d->comm = nullptr;
    d->broken = true;
}

// hipStreamSynchronize with a deadline and an eye on the communicator: returns FF_ERR_COMM (communicator abandoned) when RCCL
// reports an asynchronous error or the stream has not drained timeout_s after `local_done` (an event behind this rank's own
// work on the stream; null: from now).  The deadline bounds the wait for the OTHER ranks: while this rank's own strips are
// still rendering - 1080p at 400 000 spp takes minutes - nothing is late.
int wait_stream(FfDistContext* d, int device, hipStream_t stream, const char* what, hipEvent_t local_done = nullptr)
{
    auto t0 = std::chrono::steady_clock::now();
    const auto t_begin = t0;
    bool local_pending = local_done != nullptr;
    for (;;) {
        if (local_pending) {
            const hipError_t lq = hipEventQuery(local_done);
            if (lq == hipErrorNotReady) t0 = std::chrono::steady_clock::now(); // own work still running: the clock has not started
            else local_pending = false;                                        // (drained, or an error the stream query below reports)
        }
        const hipError_t q = hipStreamQuery(stream);
        if (q == hipSuccess) return FF_OK;
        if (q != hipErrorNotReady) {
            abandon_comm(d);
            return fail(FF_ERR_HIP, "rank %d (device %d): %s failed: %s", d->rank, device, what, hipGetErrorString(q));
        }
        if (g_rccl.CommGetAsyncError && d->comm) {
            ncclResult_t ar = ncclSuccess;
            if (g_rccl.CommGetAsyncError(d->comm, &ar) == ncclSuccess && ar != ncclSuccess && ar != ncclInProgress) {
                abandon_comm(d);
                return fail(FF_ERR_COMM, "rank %d: RCCL reported an asynchronous error during %s: %s", d->rank, what, g_rccl.GetErrorString(ar));
            }
        }
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (waited > d->timeout_s) {
            abandon_comm(d);
            return fail(FF_ERR_COMM, "rank %d: %s did not complete within %.0f s (FF_DIST_TIMEOUT_S): a peer rank is missing or stuck; communicator aborted",
                        d->rank, what, d->timeout_s);
        }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() > 200e-6)
            std::this_thread::sleep_for(std::chrono::microseconds(50)); // (a frame renders for milliseconds: do not burn a core on it)
    }
}

} // namespace

namespace ff {

void dist_release(FfState* s)
{
    FfDistContext* d = s->dist;
    if (!d) return;
    if (d->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(d->comm);
    if (d->d_pack) (void)hipFree(d->d_pack);
    if (d->d_gather) (void)hipFree(d->d_gather);
    if (d->d_status) (void)hipFree(d->d_status);
    if (d->h_status) (void)hipHostFree(d->h_status);
    if (d->ev_local) (void)hipEventDestroy(d->ev_local);
    delete d;
    s->dist = nullptr;
}

} // namespace ff

// Several devices driven by one process.
struct FfMulti {
    std::vector<FfState*> states;
    std::vector<hipStream_t> streams;   // one non-blocking stream per state
    std::vector<ncclComm_t> comms;      // transport rccl: one communicator per state (ncclCommInitAll)
    std::vector<hipEvent_t> sent;       // transport peer: "this part's copy to device 0 has been enqueued"
    std::vector<unsigned char*> d_pack; // per state (index 0 unused: the root renders into the gather buffer)
    std::vector<size_t> pack_bytes;
    unsigned char* d_gather = nullptr;
    size_t gather_bytes = 0;
    bool use_rccl = false;
    FfStats stats;
};

extern "C" {

int ff_dist_unique_id(void* out_id, int bytes)
{
    clear_error();
    if (!out_id || bytes < (int)sizeof(ncclUniqueId)) return fail(FF_ERR_INVALID_ARG, "ff_dist_unique_id: need a buffer of %d bytes", (int)sizeof(ncclUniqueId));
    int st = load_rccl();
    if (st != FF_OK) return st;
    ncclUniqueId id;
    FF_NCCL(g_rccl.GetUniqueId(&id));
    std::memcpy(out_id, &id, sizeof id);
    return FF_OK;
}

int ff_dist_init(FfState* s, int rank, int world_size, const void* id, int bytes)
{
    clear_error();
    if (!s) return fail(FF_ERR_INVALID_ARG, "ff_dist_init: state is null");
    if (world_size < 1 || rank < 0 || rank >= world_size) return fail(FF_ERR_INVALID_ARG, "ff_dist_init: rank %d of %d", rank, world_size);
    if (!id || bytes < (int)sizeof(ncclUniqueId)) return fail(FF_ERR_INVALID_ARG, "ff_dist_init: the id must be the %d bytes ff_dist_unique_id produced on rank 0", (int)sizeof(ncclUniqueId));
    int st = load_rccl();
    if (st != FF_OK) return st;
    dist_release(s);
    FF_HIP(hipSetDevice(s->device));
    FfDistContext* d = new (std::nothrow) FfDistContext();
    if (!d) return fail(FF_ERR_OOM, "ff_dist_init: out of host memory");
    d->rank = rank;
    d->world = world_size;
    if (const char* e = std::getenv("FF_DIST_SELF_LOOP")) d->self_loop = std::atoi(e) != 0;
    if (const char* e = std::getenv("FF_DIST_TIMEOUT_S")) d->timeout_s = std::max(1.0, std::atof(e));
    if (const char* e = std::getenv("FF_DEBUG_DIST_FAIL_RANK")) d->fail_rank = std::atoi(e);
    if (hipMalloc((void**)&d->d_status, 2 * sizeof(int)) != hipSuccess || hipHostMalloc((void**)&d->h_status, 2 * sizeof(int)) != hipSuccess ||
        hipEventCreateWithFlags(&d->ev_local, hipEventDisableTiming) != hipSuccess) {
        if (d->d_status) (void)hipFree(d->d_status);
        if (d->h_status) (void)hipHostFree(d->h_status);
        delete d;
        return fail(FF_ERR_OOM, "ff_dist_init: no memory for the status words");
    }
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof uid);
    ncclResult_t r = g_rccl.CommInitRank(&d->comm, world_size, uid, rank);
    if (r != ncclSuccess) {
        (void)hipFree(d->d_status);
        (void)hipHostFree(d->h_status);
        (void)hipEventDestroy(d->ev_local);
        delete d;
        return fail(FF_ERR_COMM, "ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, world_size, s->device, g_rccl.GetErrorString(r));
    }
    s->dist = d;
    return FF_OK;
}

int ff_dist_shutdown(FfState* s)
{
    clear_error();
    if (!s) return fail(FF_ERR_INVALID_ARG, "ff_dist_shutdown: state is null");
    (void)hipSetDevice(s->device);
    dist_release(s);
    return FF_OK;
}

int ff_debug_dist_fail_rank(FfState* s, int rank)
{
    clear_error();
    if (!s || !s->dist) return fail(FF_ERR_INVALID_ARG, "ff_debug_dist_fail_rank: call ff_dist_init first");
    s->dist->fail_rank = rank;
    return FF_OK;
}

int ff_dist_strip_rows(int world_size) { return default_strip_rows(world_size < 1 ? 1 : world_size, 1080); }
int ff_dist_strip_rows_for(int height, int world_size) { return default_strip_rows(world_size < 1 ? 1 : world_size, height); }

// The wire layout of the gather (PackLayout), for callers and tests: the message part `part` sends to rank 0 and where it lands
// in rank 0's gather buffer.  A part without rows has no message: neither side posts one (the same predicate on both sides).
long long ff_dist_part_bytes(int width, int height, int strip_rows, int part, int num_parts, long long* out_offset)
{
    if (width < 0 || height < 0 || strip_rows < 1 || num_parts < 1 || part < 0 || part >= num_parts) return -1;
    PackLayout L;
    L.width = width;
    L.height = height;
    L.strip_rows = strip_rows;
    L.num_parts = num_parts;
    if (out_offset) *out_offset = (long long)L.part_offset(part);
    return (long long)L.part_bytes(part);
}

int ff_dist_available(void)
{
    clear_error();
    return load_rccl();
}

// One frame over all ranks.  No rank may be left waiting for a message that will never come, so the order is: (1) everything
// that can fail locally - arguments, buffers, the launches of this rank's strips; (2) the ranks AGREE on a status, one 4-byte
// all-reduce behind the strips on the same stream, which every rank that got this far posts whatever happened to it in (1);
// (3) only if all are well, the gather: one message per peer in one group.  A rank that failed in (1) makes the frame an
// error on EVERY rank (FF_ERR_COMM on the others) and the communicator stays usable.  Every wait on the other ranks has a
// deadline and watches ncclCommGetAsyncError; when it runs out the communicator is aborted, the call returns FF_ERR_COMM and
// so does every later one until ff_dist_init has made a new communicator (a fresh process is the retry).
int ff_render_distributed(FfState* s, const FfCamera* camera, const FfRenderParams* params, int strip_rows, void* rgb8, int rgb8_on_device,
                          float* radiance, int radiance_on_device)
{
    clear_error();
    const auto t0 = std::chrono::steady_clock::now();
    if (!s) return fail(FF_ERR_INVALID_ARG, "ff_render_distributed: state is null");
    FfDistContext* d = s->dist;
    if (!d) return fail(FF_ERR_INVALID_ARG, "ff_render_distributed: call ff_dist_init first");
    if (d->broken || !d->comm) return fail(FF_ERR_COMM, "ff_render_distributed: the communicator was aborted by an earlier failure; call ff_dist_init again");
    FF_HIP(hipSetDevice(s->device));
    const int rank = d->rank, world = d->world;
    const bool root = rank == 0;
    const bool loop_back = root && d->self_loop;

    // (1) local work
    PackLayout L;
    RootOutputs out;
    unsigned char* pack = nullptr;
    int local = check_render_call(s, camera, params, "ff_render_distributed");
    if (local == FF_OK && d->fail_rank == rank) local = fail(FF_ERR_OOM, "injected failure on rank %d (FF_DEBUG_DIST_FAIL_RANK)", rank);
    if (local == FF_OK) {
        if (strip_rows <= 0) strip_rows = default_strip_rows(world, params->height);
        L.width = params->width;
        L.height = params->height;
        L.strip_rows = strip_rows;
        L.num_parts = world;
        if (root) {
            local = root_outputs(s, params, rgb8, rgb8_on_device, radiance, radiance_on_device, out);
            if (local == FF_OK) local = ensure_bytes((void**)&d->d_gather, &d->gather_bytes, L.total_bytes() + 16);
            pack = d->d_gather; // part 0 sits at offset 0: rendered in place
        }
        if (local == FF_OK && (!root || loop_back)) {
            local = ensure_bytes((void**)&d->d_pack, &d->pack_bytes, L.part_bytes(rank) + 16);
            pack = d->d_pack;
        }
        if (local == FF_OK)
            local = render_enqueue(s, camera, params, strip_rows, rank, world, ff_strips_local_rows(L.height, strip_rows, rank, world), pack + L.rad_bytes(rank),
                                   reinterpret_cast<float*>(pack));
    }

    // (2) agreement (the error text of a local failure stays this thread's last error)
    int verdict = local;
    if (world > 1 || loop_back) {
        d->h_status[0] = local == FF_OK ? 0 : 1;
        d->h_status[1] = -1;
        hipError_t e = hipEventRecord(d->ev_local, s->stream); // behind this rank's strips: the deadline of the wait below starts there
        if (e == hipSuccess) e = hipMemcpyAsync(d->d_status, d->h_status, sizeof(int), hipMemcpyHostToDevice, s->stream);
        ncclResult_t r = ncclSuccess;
        if (e == hipSuccess) r = g_rccl.AllReduce(d->d_status, d->d_status + 1, 1, ncclInt32, ncclMax, d->comm, s->stream);
        if (e == hipSuccess && r == ncclSuccess) e = hipMemcpyAsync(d->h_status + 1, d->d_status + 1, sizeof(int), hipMemcpyDeviceToHost, s->stream);
        if (e != hipSuccess || r != ncclSuccess) {
            // this rank cannot even post its status: the others will run into their deadline; nothing more to do here
            abandon_comm(d);
            return fail(FF_ERR_COMM, "rank %d could not post its frame status: %s", rank, e != hipSuccess ? hipGetErrorString(e) : g_rccl.GetErrorString(r));
        }
        const int wst = wait_stream(d, s->device, s->stream, "the ranks' status agreement", d->ev_local);
        if (wst != FF_OK) return wst;
        if (d->h_status[1] != 0 && local == FF_OK) verdict = FF_ERR_COMM;
    }
    if (verdict != FF_OK) {
        // the frame is off on every rank; what this rank had enqueued has drained (or never started)
        if (local == FF_OK) {
            (void)render_finish(s);
            return fail(FF_ERR_COMM, "ff_render_distributed: another rank failed before the gather (its own call reports why); frame dropped on rank %d", rank);
        }
        s->pending = false;
        return local;
    }

    // (3) the gather: one message per peer, all in one group, on the stream the strips were rendered on
    const auto t_gather = std::chrono::steady_clock::now();
    if (world > 1 || loop_back) {
        ncclResult_t r = g_rccl.GroupStart();
        if (r == ncclSuccess && (!root || loop_back) && L.part_bytes(rank) > 0) r = g_rccl.Send(pack, L.part_bytes(rank), ncclChar, 0, d->comm, s->stream);
        if (root && r == ncclSuccess) {
            for (int p = loop_back ? 0 : 1; p < world && r == ncclSuccess; ++p)
                if (L.part_bytes(p) > 0) r = g_rccl.Recv(d->d_gather + L.part_offset(p), L.part_bytes(p), ncclChar, p, d->comm, s->stream);
        }
        const ncclResult_t e = g_rccl.GroupEnd();
        if (r != ncclSuccess || e != ncclSuccess) {
            abandon_comm(d);
            return fail(FF_ERR_COMM, "framebuffer gather failed on rank %d: %s; communicator aborted", rank, g_rccl.GetErrorString(r != ncclSuccess ? r : e));
        }
    }
    if (root) FF_HIP(launch_unpack_strips(d->d_gather, out.rgb8, out.radiance, L.width, L.height, strip_rows, world, s->stream));
    int st = wait_stream(d, s->device, s->stream, "the framebuffer gather"); // (a rank without rows enqueued no frame but still took part in the gather)
    if (st != FF_OK) return st;
    st = render_finish(s);
    if (st != FF_OK) return st;
    d->last_gather_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_gather).count();
    if (root) {
        st = copy_root_outputs_to_host(s, params, rgb8, rgb8_on_device, radiance, radiance_on_device, out);
        if (st != FF_OK) return st;
    }
    s->stats.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return FF_OK;
}

// ---- one process, several devices ----------------------------------------------------------------------------------

int ff_multi_destroy(FfMulti* m)
{
    if (!m) return FF_OK;
    for (size_t i = 0; i < m->states.size(); ++i) {
        FfState* s = m->states[i];
        if (!s) continue;
        (void)hipSetDevice(s->device);
        if (i < m->comms.size() && m->comms[i] && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(m->comms[i]);
        if (i < m->d_pack.size() && m->d_pack[i]) (void)hipFree(m->d_pack[i]);
        if (i < m->sent.size() && m->sent[i]) (void)hipEventDestroy(m->sent[i]);
        if (i == 0 && m->d_gather) (void)hipFree(m->d_gather);
        s->stream = nullptr;
        if (i < m->streams.size() && m->streams[i]) (void)hipStreamDestroy(m->streams[i]);
        ff_destroy(s);
    }
    delete m;
    return FF_OK;
}

int ff_multi_create(FfMulti** out, const int* device_ids, int n)
{
    clear_error();
    if (!out || !device_ids || n < 1 || n > 64) return fail(FF_ERR_INVALID_ARG, "ff_multi_create: need 1..64 device ids");
    *out = nullptr;
    FfMulti* m = new (std::nothrow) FfMulti();
    if (!m) return fail(FF_ERR_OOM, "ff_multi_create: out of host memory");
    m->states.assign(n, nullptr);
    m->streams.assign(n, nullptr);
    m->comms.assign(n, nullptr);
    m->sent.assign(n, nullptr);
    m->d_pack.assign(n, nullptr);
    m->pack_bytes.assign(n, 0);
    bool distinct = true;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j) distinct = distinct && device_ids[i] != device_ids[j];
    for (int i = 0; i < n; ++i) {
        int st = ff_create(&m->states[i], device_ids[i]);
        if (st == FF_OK && hipStreamCreateWithFlags(&m->streams[i], hipStreamNonBlocking) != hipSuccess) st = fail(FF_ERR_HIP, "ff_multi_create: stream creation failed on device %d", device_ids[i]);
        if (st == FF_OK && hipEventCreateWithFlags(&m->sent[i], hipEventDisableTiming) != hipSuccess) st = fail(FF_ERR_HIP, "ff_multi_create: event creation failed on device %d", device_ids[i]);
        if (st != FF_OK) {
            ff_multi_destroy(m);
            return st;
        }
        m->states[i]->stream = m->streams[i];
    }
    const char* forced = std::getenv("FF_MULTI_TRANSPORT");
    m->use_rccl = n > 1 && distinct && !(forced && std::strcmp(forced, "peer") == 0);
    if (m->use_rccl) {
        int st = load_rccl();
        if (st == FF_OK) {
            ncclResult_t r = g_rccl.CommInitAll(m->comms.data(), n, device_ids);
            if (r != ncclSuccess) st = fail(FF_ERR_COMM, "ncclCommInitAll over %d devices failed: %s", n, g_rccl.GetErrorString(r));
        }
        if (st != FF_OK) {
            ff_multi_destroy(m);
            return st;
        }
    } else if (n > 1) {
        // peer copies into device 0's gather buffer
        for (int i = 1; i < n; ++i) {
            if (device_ids[i] == device_ids[0]) continue;
            (void)hipSetDevice(device_ids[i]);
            hipError_t e = hipDeviceEnablePeerAccess(device_ids[0], 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError(); // copies still work, staged by the runtime
        }
    }
    *out = m;
    return FF_OK;
}

int ff_multi_count(const FfMulti* m) { return m ? (int)m->states.size() : 0; }

FfState* ff_multi_state(FfMulti* m, int index)
{
    if (!m || index < 0 || index >= (int)m->states.size()) return nullptr;
    return m->states[index];
}

int ff_multi_uses_rccl(const FfMulti* m) { return m && m->use_rccl ? 1 : 0; }

int ff_multi_upload_scene(FfMulti* m, const FfGeometry* host_geometries, int n)
{
    clear_error();
    if (!m) return fail(FF_ERR_INVALID_ARG, "ff_multi_upload_scene: handle is null");
    // The scene is replicated: every GPU traces against all of it.  With the host builder it is compiled ONCE (records, SAH
    // trees, triangle records: the expensive part, 436 ms for the 983 040-triangle sphere) and the compiled arrays go to every
    // device (kernel.cu:268-298 uploads once, too).  Device builders run per device: their builds take milliseconds.
    bool all_host = true;
    for (FfState* s : m->states) all_host = all_host && s->builder == FF_BUILD_HOST_SAH;
    if (all_host && m->states.size() > 1) {
        CompiledScene cs;
        const auto t0 = std::chrono::steady_clock::now();
        const int st = compile_scene(host_geometries, n, default_bvh_params(), cs);
        if (st != FF_OK) return st;
        const double build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        for (size_t i = 0; i < m->states.size(); ++i) {
            const int ust = upload_compiled_scene(m->states[i], cs, i == 0 ? build_ms : 0.0); // (the build is charged to the first state)
            if (ust != FF_OK) return ust;
        }
        return FF_OK;
    }
    for (FfState* s : m->states) {
        const int st = ff_upload_scene(s, host_geometries, n);
        if (st != FF_OK) return st;
    }
    return FF_OK;
}

namespace {

// The frame on all devices of `m`; final outputs (device pointers on states[0]'s device, either may be null).
int multi_render_core(FfMulti* m, const FfCamera* camera, const FfRenderParams* params, int strip_rows, unsigned char* rgb8_dev, float* radiance_dev)
{
    const int n = (int)m->states.size();
    FfState* root = m->states[0];
    m->stats = FfStats();
    if (n == 1) {
        FF_HIP(hipSetDevice(root->device));
        int st = render_enqueue(root, camera, params, params->height, 0, 1, params->height, rgb8_dev, radiance_dev);
        if (st == FF_OK) st = render_finish(root);
        m->stats = root->stats;
        return st;
    }
    if (strip_rows <= 0) strip_rows = default_strip_rows(n, params->height);
    PackLayout L;
    L.width = params->width;
    L.height = params->height;
    L.strip_rows = strip_rows;
    L.num_parts = n;
    FF_HIP(hipSetDevice(root->device));
    int st = ensure_bytes((void**)&m->d_gather, &m->gather_bytes, L.total_bytes() + 16);
    if (st != FF_OK) return st;
    // 1. every device starts on its strips (nothing here waits for a GPU)
    for (int i = 0; i < n; ++i) {
        FfState* s = m->states[i];
        FF_HIP(hipSetDevice(s->device));
        unsigned char* pack = m->d_gather; // the root renders part 0 in place
        if (i > 0) {
            st = ensure_bytes((void**)&m->d_pack[i], &m->pack_bytes[i], L.part_bytes(i) + 16);
            if (st != FF_OK) return st;
            pack = m->d_pack[i];
        }
        st = render_enqueue(s, camera, params, strip_rows, i, n, ff_strips_local_rows(L.height, strip_rows, i, n), pack + L.rad_bytes(i),
                            reinterpret_cast<float*>(pack));
        if (st != FF_OK) return st;
    }
    // 2. the gather, behind each device's own rendering
    if (m->use_rccl) {
        FF_NCCL(g_rccl.GroupStart());
        ncclResult_t r = ncclSuccess;
        for (int i = 1; i < n && r == ncclSuccess; ++i) {
            if (L.part_bytes(i) == 0) continue;
            r = g_rccl.Send(m->d_pack[i], L.part_bytes(i), ncclChar, 0, m->comms[i], m->streams[i]);
            if (r == ncclSuccess) r = g_rccl.Recv(m->d_gather + L.part_offset(i), L.part_bytes(i), ncclChar, i, m->comms[0], m->streams[0]);
        }
        const ncclResult_t e = g_rccl.GroupEnd();
        if (r != ncclSuccess || e != ncclSuccess) return fail(FF_ERR_COMM, "framebuffer gather failed: %s", g_rccl.GetErrorString(r != ncclSuccess ? r : e));
    } else {
        for (int i = 1; i < n; ++i) {
            if (L.part_bytes(i) == 0) continue;
            FfState* s = m->states[i];
            FF_HIP(hipSetDevice(s->device));
            FF_HIP(hipMemcpyPeerAsync(m->d_gather + L.part_offset(i), root->device, m->d_pack[i], s->device, L.part_bytes(i), m->streams[i]));
            FF_HIP(hipEventRecord(m->sent[i], m->streams[i]));
            FF_HIP(hipSetDevice(root->device));
            FF_HIP(hipStreamWaitEvent(m->streams[0], m->sent[i], 0));
        }
    }
    // 3. strips -> image order on the root, then wait for everybody
    FF_HIP(hipSetDevice(root->device));
    FF_HIP(launch_unpack_strips(m->d_gather, rgb8_dev, radiance_dev, L.width, L.height, strip_rows, n, m->streams[0]));
    for (int i = n - 1; i >= 0; --i) { // the root last: its stream carries the scatter
        FfState* s = m->states[i];
        FF_HIP(hipSetDevice(s->device));
        st = render_finish(s);
        if (st != FF_OK) return st;
        if (i == 0) FF_HIP(hipStreamSynchronize(m->streams[0]));
        m->stats.rays_traced += s->stats.rays_traced;
        m->stats.rays_answered += s->stats.rays_answered;
        m->stats.rays_cut_short += s->stats.rays_cut_short;
        m->stats.nodes_visited += s->stats.nodes_visited;
        m->stats.tris_tested += s->stats.tris_tested;
        m->stats.planes_tested += s->stats.planes_tested;
        m->stats.kernel_ms = std::max(m->stats.kernel_ms, s->stats.kernel_ms);
        m->stats.kernel_launches = std::max(m->stats.kernel_launches, s->stats.kernel_launches);
        m->stats.scene_bytes_nodes = s->stats.scene_bytes_nodes;
        m->stats.scene_bytes_tris = s->stats.scene_bytes_tris;
    }
    FF_HIP(hipSetDevice(root->device));
    return FF_OK;
}

} // namespace

int ff_multi_render(FfMulti* m, const FfCamera* camera, const FfRenderParams* params, int strip_rows, void* rgb8, int rgb8_on_device, float* radiance,
                    int radiance_on_device)
{
    clear_error();
    const auto t0 = std::chrono::steady_clock::now();
    if (!m) return fail(FF_ERR_INVALID_ARG, "ff_multi_render: handle is null");
    for (FfState* s : m->states) {
        const int st = check_render_call(s, camera, params, "ff_multi_render");
        if (st != FF_OK) return st;
    }
    FfState* root = m->states[0];
    FF_HIP(hipSetDevice(root->device));
    RootOutputs out;
    int st = root_outputs(root, params, rgb8, rgb8_on_device, radiance, radiance_on_device, out);
    if (st != FF_OK) return st;
    st = multi_render_core(m, camera, params, strip_rows, out.rgb8, out.radiance);
    if (st != FF_OK) return st;
    st = copy_root_outputs_to_host(root, params, rgb8, rgb8_on_device, radiance, radiance_on_device, out);
    if (st != FF_OK) return st;
    m->stats.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return FF_OK;
}

int ff_multi_render_to_pbo(FfMulti* m, const FfCamera* camera, const FfRenderParams* params, int strip_rows)
{
    clear_error();
    const auto t0 = std::chrono::steady_clock::now();
    if (!m) return fail(FF_ERR_INVALID_ARG, "ff_multi_render_to_pbo: handle is null");
    FfState* root = m->states[0];
    if (!root->pbo_resource) return fail(FF_ERR_GL_UNAVAILABLE, "ff_multi_render_to_pbo: no pixel buffer registered on device 0's state (ff_register_gl_pbo(ff_multi_state(m, 0), ...))");
    for (FfState* s : m->states) {
        const int st = check_render_call(s, camera, params, "ff_multi_render_to_pbo");
        if (st != FF_OK) return st;
    }
    if (params->width != root->pbo_width || params->height != root->pbo_height)
        return fail(FF_ERR_INVALID_ARG, "ff_multi_render_to_pbo: params are %dx%d but the registered buffer is %dx%d", params->width, params->height, root->pbo_width, root->pbo_height);
    FF_HIP(hipSetDevice(root->device));
    // kernel.cu:335-344 with the kernel replaced by "every GPU renders its strips, device 0 gathers into the mapped buffer"
    void* dptr = nullptr;
    size_t nbytes = 0;
    FF_HIP(hipGraphicsMapResources(1, &root->pbo_resource, root->stream));
    hipError_t e = hipGraphicsResourceGetMappedPointer(&dptr, &nbytes, root->pbo_resource);
    if (e == hipSuccess && nbytes < (size_t)params->width * (size_t)params->height * 3) e = hipErrorInvalidValue;
    int st;
    if (e == hipSuccess) st = multi_render_core(m, camera, params, strip_rows, (unsigned char*)dptr, nullptr);
    else st = fail(FF_ERR_HIP, "mapping the pixel buffer failed: %s", hipGetErrorString(e));
    (void)hipSetDevice(root->device);
    hipError_t ue = hipGraphicsUnmapResources(1, &root->pbo_resource, root->stream);
    if (st == FF_OK && ue != hipSuccess) st = fail(FF_ERR_HIP, "hipGraphicsUnmapResources failed: %s", hipGetErrorString(ue));
    m->stats.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return st;
}

int ff_multi_stats(FfMulti* m, FfStats* out)
{
    clear_error();
    if (!m || !out) return fail(FF_ERR_INVALID_ARG, "ff_multi_stats: null argument");
    *out = m->stats;
    return FF_OK;
}

} // extern "C"
