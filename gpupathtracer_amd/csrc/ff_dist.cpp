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
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
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
    FF_SYM(GetErrorString, "ncclGetErrorString")
#undef FF_SYM
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

int default_strip_rows(int world)
{
    // thin strips for many ranks: every rank then holds the same number of rows to within one strip, and neighbouring
    // strips (similar cost) go to different ranks (1080 rows over 8 ranks: 4-row strips -> 136 or 132 rows per rank)
    return world <= 2 ? 16 : (world <= 4 ? 8 : 4);
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
    unsigned char* d_pack = nullptr; // this rank's packed strips (non-root ranks; rank 0 with self_loop)
    size_t pack_bytes = 0;
    unsigned char* d_gather = nullptr; // rank 0: every part's packed strips, part after part
    size_t gather_bytes = 0;
    double last_gather_ms = 0.0;
};

namespace ff {

void dist_release(FfState* s)
{
    FfDistContext* d = s->dist;
    if (!d) return;
    if (d->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(d->comm);
    if (d->d_pack) (void)hipFree(d->d_pack);
    if (d->d_gather) (void)hipFree(d->d_gather);
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
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof uid);
    ncclResult_t r = g_rccl.CommInitRank(&d->comm, world_size, uid, rank);
    if (r != ncclSuccess) {
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

int ff_dist_strip_rows(int world_size) { return default_strip_rows(world_size < 1 ? 1 : world_size); }

int ff_render_distributed(FfState* s, const FfCamera* camera, const FfRenderParams* params, int strip_rows, void* rgb8, int rgb8_on_device,
                          float* radiance, int radiance_on_device)
{
    clear_error();
    const auto t0 = std::chrono::steady_clock::now();
    int st = check_render_call(s, camera, params, "ff_render_distributed");
    if (st != FF_OK) return st;
    FfDistContext* d = s->dist;
    if (!d) return fail(FF_ERR_INVALID_ARG, "ff_render_distributed: call ff_dist_init first");
    if (strip_rows <= 0) strip_rows = default_strip_rows(d->world);
    FF_HIP(hipSetDevice(s->device));
    PackLayout L;
    L.width = params->width;
    L.height = params->height;
    L.strip_rows = strip_rows;
    L.num_parts = d->world;
    const int rank = d->rank, world = d->world;
    const int local_rows = ff_strips_local_rows(L.height, strip_rows, rank, world);
    const bool root = rank == 0;
    const bool loop_back = root && d->self_loop;

    RootOutputs out;
    unsigned char* pack = nullptr;
    if (root) {
        st = root_outputs(s, params, rgb8, rgb8_on_device, radiance, radiance_on_device, out);
        if (st == FF_OK) st = ensure_bytes((void**)&d->d_gather, &d->gather_bytes, L.total_bytes() + 16);
        if (st != FF_OK) return st;
        pack = d->d_gather; // part 0 sits at offset 0: rendered in place
    }
    if (!root || loop_back) {
        st = ensure_bytes((void**)&d->d_pack, &d->pack_bytes, L.part_bytes(rank) + 16);
        if (st != FF_OK) return st;
        pack = d->d_pack;
    }
    st = render_enqueue(s, camera, params, strip_rows, rank, world, local_rows, pack + L.rad_bytes(rank), reinterpret_cast<float*>(pack));
    if (st != FF_OK) return st;

    // The gather: one message per peer, all in one group, on the stream the strips were rendered on.
    const auto t_gather = std::chrono::steady_clock::now();
    if (world > 1 || loop_back) {
        FF_NCCL(g_rccl.GroupStart());
        ncclResult_t r = ncclSuccess;
        if (!root || loop_back) {
            if (L.part_bytes(rank) > 0) r = g_rccl.Send(pack, L.part_bytes(rank), ncclChar, 0, d->comm, s->stream);
        }
        if (root && r == ncclSuccess) {
            for (int p = loop_back ? 0 : 1; p < world && r == ncclSuccess; ++p)
                if (L.part_bytes(p) > 0) r = g_rccl.Recv(d->d_gather + L.part_offset(p), L.part_bytes(p), ncclChar, p, d->comm, s->stream);
        }
        const ncclResult_t e = g_rccl.GroupEnd();
        if (r != ncclSuccess || e != ncclSuccess)
            return fail(FF_ERR_COMM, "framebuffer gather failed on rank %d: %s", rank, g_rccl.GetErrorString(r != ncclSuccess ? r : e));
    }
    if (root) FF_HIP(launch_unpack_strips(d->d_gather, out.rgb8, out.radiance, L.width, L.height, strip_rows, world, s->stream));
    st = render_finish(s);
    if (st != FF_OK) return st;
    FF_HIP(hipStreamSynchronize(s->stream)); // (a rank without rows enqueued no frame but still took part in the gather)
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
    for (FfState* s : m->states) {
        const int st = ff_upload_scene(s, host_geometries, n); // the scene is replicated: every GPU traces against all of it
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
    if (strip_rows <= 0) strip_rows = default_strip_rows(n);
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
