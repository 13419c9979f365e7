// Several GPUs of one node behind one handle: the rt_group_* entry points of
// rt_mi355.h.  The reference's seam is a single `Scene::render(&self)` call
// (/root/reference/src/lib.rs:188); a host that hands that call to this library
// with a list of GPUs gets the image one GPU would store, bit for bit.
//
// The scene is replicated (one context per listed GPU).  A pass's samples are
// dealt out in contiguous pieces, every member traces its piece at the same time
// on a host thread of its own (rt_trace_samples: per-sample radiance stays on
// the member's GPU), then the running per-pixel sums travel from member to
// member in sample order (hipMemcpyPeerAsync on the receiver's stream) and each member folds its samples onto
// them (rt_fold_samples): the left fold of lib.rs:274-279 is the same sequence
// of additions whatever the number of members.  No collective is needed: the
// path has no exchange step other than this hand-over of 12 bytes per pixel.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <new>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "abi_guard.h"
#include "rt_mi355.h"

struct RtGroup {
    std::vector<RtContext*> member;
    std::vector<int> device;
    std::vector<float*> d_sums;  // per member, on its GPU: 3 floats per pixel
    size_t cap_pixels = 0;
    std::string error;
};

// rt_context.hip: makes the context's GPU current and hands out the stream its work is enqueued on
int rt_context_device(RtContext* ctx, hipStream_t* stream);

namespace {

thread_local char g_group_error[512] = "";

// Worker threads that are joined whichever way their scope is left: a thread that could not be
// started (std::system_error) must not leave its already running siblings to be destroyed unjoined.
struct Workers {
    std::vector<std::thread> threads;
    template <class... A>
    void start(A&&... a) { threads.emplace_back(std::forward<A>(a)...); }
    void join() {
        for (std::thread& t : threads)
            if (t.joinable()) t.join();
    }
    ~Workers() { join(); }
};

int group_fail(RtGroup* g, int code, const std::string& msg) {
    g->error = msg;
    return code;
}

// A member's failure, with its message.
int member_fail(RtGroup* g, size_t k, int code) {
    g->error = "GPU " + std::to_string(g->device[k]) + " (member " + std::to_string(k) + "): " + rt_last_error(g->member[k]);
    return code;
}

void free_sums(RtGroup* g) {
    for (size_t k = 0; k < g->d_sums.size(); ++k) {
        if (!g->d_sums[k]) continue;
        (void)hipSetDevice(g->device[k]);
        (void)hipFree(g->d_sums[k]);
        g->d_sums[k] = nullptr;
    }
    g->cap_pixels = 0;
}

int ensure_sums(RtGroup* g, size_t n_pixels) {
    if (g->cap_pixels >= n_pixels) return RT_OK;
    free_sums(g);
    for (size_t k = 0; k < g->member.size(); ++k) {
        if (hipSetDevice(g->device[k]) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&g->d_sums[k]), n_pixels * 3 * sizeof(float)) != hipSuccess) {
            (void)hipGetLastError();
            free_sums(g);
            return group_fail(g, RT_ERR_OOM, "no memory for the per-pixel sums on GPU " + std::to_string(g->device[k]));
        }
    }
    g->cap_pixels = n_pixels;
    return RT_OK;
}

size_t pixels_of(const RtRenderParams* p) {
    return p->tile_w ? static_cast<size_t>(p->tile_w) * p->tile_h : static_cast<size_t>(p->width) * p->height;
}

// Samples [begin, begin + count) of every pixel folded onto the sums of the samples before
// them (`carry_on`: member 0's buffer holds those; otherwise the fold starts empty).  On
// return the buffer of the last member that had samples holds the result; its index is
// returned in *holder.
//
// The samples are taken in rounds of one piece per member — member k the k-th piece of the
// round, all pieces traced at the same time, then folded in member order — with pieces no
// larger than the smallest number of samples any member holds in flight, so that no GPU
// waits while another works through a share that did not fit at once.
int group_sums(RtGroup* g, const RtRenderParams* p, uint32_t begin, uint32_t count, bool carry_on, size_t* holder) {
    const uint32_t n = static_cast<uint32_t>(g->member.size());
    const size_t bytes = pixels_of(p) * 3 * sizeof(float);
    uint32_t fit = (count + n - 1) / n;  // the whole pass in one round, if it fits
    {
        const uint32_t want = fit;
        for (uint32_t k = 0; k < n && k < count; ++k) {
            uint32_t here = 0;
            const int rc = rt_reserve_samples(g->member[k], p, want, &here);
            if (rc != RT_OK) return member_fail(g, k, rc);
            fit = std::min(fit, here);
        }
    }
    std::vector<uint32_t> first(n), mine(n), traced(n);
    std::vector<int> rc(n);
    size_t last = 0;
    bool have = carry_on;  // does some buffer hold the fold so far?
    for (uint32_t done = 0; done < count;) {
        const uint32_t left = count - done;
        const uint32_t piece = std::min(fit, (left + n - 1) / n);
        for (uint32_t k = 0; k < n; ++k) {
            const uint64_t at = static_cast<uint64_t>(k) * piece;
            first[k] = done + static_cast<uint32_t>(std::min<uint64_t>(at, left));
            mine[k] = at < left ? std::min<uint32_t>(piece, left - static_cast<uint32_t>(at)) : 0u;
            traced[k] = 0;
            rc[k] = RT_OK;
        }
        {
            Workers workers;
            for (uint32_t k = 1; k < n; ++k)
                if (mine[k])
                    workers.start([&, k] { rc[k] = rt_trace_samples(g->member[k], p, begin + first[k], mine[k], &traced[k]); });
            rc[0] = rt_trace_samples(g->member[0], p, begin + first[0], mine[0], &traced[0]);  // mine[0] >= 1
            workers.join();
        }
        for (uint32_t k = 0; k < n; ++k) {
            if (rc[k] != RT_OK) return member_fail(g, k, rc[k]);
            if (traced[k] != mine[k]) return group_fail(g, RT_ERR_STATE, "a member traced fewer samples than it had reserved room for");
        }
        // the sums go from member to member in sample order
        for (uint32_t k = 0; k < n; ++k) {
            if (!mine[k]) continue;
            if (have && k != last) {
                // On the receiving member's own stream: rt_fold_samples enqueues its work there, so
                // the fold is ordered after the copy (a copy between devices on the null stream is not
                // ordered against the members' non-blocking streams, and the host does not wait for
                // it either).  The sender's buffer is complete: its rt_fold_samples returned after
                // draining its stream.
                hipStream_t into = nullptr;
                const int rs = rt_context_device(g->member[k], &into);  // makes the receiver's GPU current
                if (rs != RT_OK) return member_fail(g, k, rs);
                if (hipMemcpyPeerAsync(g->d_sums[k], g->device[k], g->d_sums[last], g->device[last], bytes, into) != hipSuccess) {
                    (void)hipGetLastError();
                    return group_fail(g, RT_ERR_HIP, "copy of the sums between GPUs failed");
                }
            }
            const int r = rt_fold_samples(g->member[k], p, have ? 1 : 0, g->d_sums[k], 1);
            if (r != RT_OK) return member_fail(g, k, r);
            have = true;
            last = k;
            done += mine[k];
        }
    }
    *holder = last;
    return RT_OK;
}

// A render of many short passes over several GPUs: the region's rows are cut into bands (four per member),
// the passes are taken in groups, and within a group the members pull bands off a shared counter and render
// the group's passes of a band with one rt_render call each (which traces those passes' paths together) —
// `continue` semantics, straight into the band's rows of the caller's buffer.  No data moves between GPUs.
// A band of a pass is a tile of a pass, whose bytes are those of the untiled render; after every group of
// passes the caller's buffer is a whole image of that many passes, and progress is reported for each of them.
int group_render_bands(RtGroup* g, const RtRenderParams* p, uint8_t* rgb8, RtProgressFn progress, void* user, uint32_t n_passes,
                       uint32_t chunk, uint32_t first_pass) {
    const uint32_t n = static_cast<uint32_t>(g->member.size());
    const uint32_t x0 = p->tile_w ? p->tile_x : 0u, y0 = p->tile_w ? p->tile_y : 0u;
    const uint32_t w = p->tile_w ? p->tile_w : p->width, h = p->tile_w ? p->tile_h : p->height;
    const uint32_t n_bands = std::min(h, 4u * n);
    // passes per group: a band's call should be a batch of about 2^25 paths
    const uint64_t band_pass_paths = std::max<uint64_t>(1, static_cast<uint64_t>(chunk) * w * ((h + n_bands - 1) / n_bands));
    const uint32_t per_group = static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>(n_passes, (uint64_t{1} << 25) / band_pass_paths)));
    if ((static_cast<uint64_t>(first_pass) + n_passes) * chunk > 0xFFFFFFFFull) return group_fail(g, RT_ERR_INVALID, "sample index overflow");
    for (uint32_t pass = 0; pass < n_passes; pass += per_group) {
        const uint32_t now = std::min(per_group, n_passes - pass);
        RtRenderParams bp = *p;
        bp.samples_per_pixel = now * chunk;
        bp.incremental = chunk;
        bp.has_continue = 1;
        bp.continue_sampling = first_pass + pass;
        bp.tile_x = x0;
        bp.tile_w = w;
        std::atomic<uint32_t> next{0};
        std::vector<int> rc(n, RT_OK);
        auto work = [&](uint32_t k) {
            for (;;) {
                const uint32_t b = next.fetch_add(1);
                if (b >= n_bands || rc[k] != RT_OK) return;
                const uint32_t r0 = static_cast<uint32_t>(static_cast<uint64_t>(h) * b / n_bands);
                const uint32_t r1 = static_cast<uint32_t>(static_cast<uint64_t>(h) * (b + 1) / n_bands);
                RtRenderParams mine = bp;
                mine.tile_y = y0 + r0;
                mine.tile_h = r1 - r0;
                rc[k] = rt_render(g->member[k], &mine, rgb8 + static_cast<size_t>(r0) * w * 3, nullptr, nullptr);
            }
        };
        {
            Workers workers;
            for (uint32_t k = 1; k < n; ++k) workers.start(work, k);
            work(0);
            workers.join();
        }
        for (uint32_t k = 0; k < n; ++k)
            if (rc[k] != RT_OK) return member_fail(g, k, rc[k]);
        for (uint32_t j = 0; j < now; ++j)
            if (progress) progress(first_pass + (pass + j + 1) * chunk, user);  // lib.rs:238-246
    }
    return RT_OK;
}

int check_group(RtGroup* g, const RtRenderParams* p) {
    if (!g) return RT_ERR_INVALID;
    if (!p) return group_fail(g, RT_ERR_INVALID, "null render parameters");
    if (p->width < 2 || p->height < 2) return group_fail(g, RT_ERR_INVALID, "width and height must be at least 2");
    if ((p->tile_w != 0) != (p->tile_h != 0) || static_cast<uint64_t>(p->tile_x) + p->tile_w > p->width ||
        static_cast<uint64_t>(p->tile_y) + p->tile_h > p->height)
        return group_fail(g, RT_ERR_INVALID, "tile outside the frame");
    return ensure_sums(g, pixels_of(p));
}

}  // namespace

int rt_abi_caught(RtGroup* g, bool out_of_memory) noexcept {
    try {
        if (g) g->error.assign(out_of_memory ? "out of host memory" : "internal error (an exception was caught at the C boundary)");
    } catch (...) {
    }
    return out_of_memory ? RT_ERR_OOM : RT_ERR_HIP;
}

extern "C" {

int rt_group_create(const int* device_ordinals, uint32_t n_devices, RtGroup** out) try {
    if (!out) return RT_ERR_INVALID;
    *out = nullptr;
    g_group_error[0] = 0;
    if (!device_ordinals || n_devices == 0 || n_devices > 64) {
        std::snprintf(g_group_error, sizeof g_group_error, "a group takes a list of 1..64 GPU ordinals");
        return RT_ERR_INVALID;
    }
    RtGroup* g = new (std::nothrow) RtGroup;
    if (!g) return RT_ERR_OOM;
    for (uint32_t k = 0; k < n_devices; ++k) {
        RtContext* ctx = nullptr;
        const int rc = rt_create(device_ordinals[k], &ctx);
        if (rc != RT_OK) {
            std::snprintf(g_group_error, sizeof g_group_error, "GPU %d: %s", device_ordinals[k],
                          rc == RT_ERR_NO_DEVICE ? "no such device" : rt_last_error(nullptr));
            rt_group_destroy(g);
            return rc;
        }
        g->member.push_back(ctx);
        g->device.push_back(device_ordinals[k]);
        g->d_sums.push_back(nullptr);
    }
    // direct copies between the members' GPUs where the hardware allows them (xGMI);
    // the copies go through host memory otherwise
    for (uint32_t a = 0; a < n_devices; ++a)
        for (uint32_t b = 0; b < n_devices; ++b) {
            int can = 0;
            if (g->device[a] == g->device[b]) continue;
            if (hipDeviceCanAccessPeer(&can, g->device[a], g->device[b]) == hipSuccess && can && hipSetDevice(g->device[a]) == hipSuccess)
                (void)hipDeviceEnablePeerAccess(g->device[b], 0);  // "already enabled" is fine
            (void)hipGetLastError();
        }
    *out = g;
    return RT_OK;
} RT_ABI_END_INT

void rt_group_destroy(RtGroup* g) try {
    if (!g) return;
    free_sums(g);
    for (RtContext* ctx : g->member) rt_destroy(ctx);
    delete g;
} RT_ABI_END_VOID

const char* rt_group_last_error(const RtGroup* g) try {
    if (g) return g->error.c_str();
    return g_group_error[0] ? g_group_error : "null group";
} RT_ABI_END_VALUE("internal error")

uint32_t rt_group_size(const RtGroup* g) try { return g ? static_cast<uint32_t>(g->member.size()) : 0u; } RT_ABI_END_VALUE(0u)

RtContext* rt_group_member(RtGroup* g, uint32_t index) try { return g && index < g->member.size() ? g->member[index] : nullptr; } RT_ABI_END_VALUE(nullptr)

int rt_group_set_option(RtGroup* g, uint32_t option, uint64_t value) try {
    if (!g) return RT_ERR_INVALID;
    for (size_t k = 0; k < g->member.size(); ++k) {
        const int rc = rt_set_option(g->member[k], option, value);
        if (rc != RT_OK) return member_fail(g, k, rc);
    }
    return RT_OK;
} RT_ABI_END_CTX(g)

int rt_group_upload_scene(RtGroup* g, const RtSceneDesc* scene) try {
    if (!g) return RT_ERR_INVALID;
    const size_t n = g->member.size();
    std::vector<int> rc(n, RT_OK);
    Workers workers;
    for (size_t k = 1; k < n; ++k) workers.start([&, k] { rc[k] = rt_upload_scene(g->member[k], scene); });
    rc[0] = rt_upload_scene(g->member[0], scene);
    workers.join();
    for (size_t k = 0; k < n; ++k)
        if (rc[k] != RT_OK) return member_fail(g, k, rc[k]);
    return RT_OK;
} RT_ABI_END_CTX(g)

int rt_group_render_sums(RtGroup* g, const RtRenderParams* p, uint32_t sample_begin, uint32_t sample_count, float* sums) try {
    int rc = check_group(g, p);
    if (rc != RT_OK) return rc;
    if (!sums || sample_count == 0) return group_fail(g, RT_ERR_INVALID, "bad arguments");
    size_t holder = 0;
    rc = group_sums(g, p, sample_begin, sample_count, false, &holder);
    if (rc != RT_OK) return rc;
    if (hipSetDevice(g->device[holder]) != hipSuccess ||
        hipMemcpy(sums, g->d_sums[holder], pixels_of(p) * 3 * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        return group_fail(g, RT_ERR_HIP, "copy of the sums to the host failed");
    }
    return RT_OK;
} RT_ABI_END_CTX(g)

int rt_group_render(RtGroup* g, const RtRenderParams* p, uint8_t* rgb8, RtProgressFn progress, void* user) try {
    int rc = check_group(g, p);
    if (rc != RT_OK) return rc;
    if (!rgb8) return group_fail(g, RT_ERR_INVALID, "null pixel buffer");
    if (g->member.size() == 1) {
        // one GPU: the single-GPU call itself (which traces several short passes together)
        rc = rt_render(g->member[0], p, rgb8, progress, user);
        return rc == RT_OK ? RT_OK : member_fail(g, 0, rc);
    }
    // lib.rs:206-211
    const uint32_t chunk = p->incremental ? p->incremental : p->samples_per_pixel;
    if (chunk == 0 || p->samples_per_pixel % chunk != 0)
        return group_fail(g, RT_ERR_INVALID, "incremental must be non-zero and divide the samples per pixel");
    const uint32_t n_passes = p->samples_per_pixel / chunk;
    const int running = p->incremental != 0 || p->has_continue != 0;  // lib.rs:286
    const uint32_t first_pass = p->has_continue ? p->continue_sampling : 0;
    // Many short passes (`incremental(1)` is what the reference's scenes ship with): a pass's samples cannot be
    // shared out usefully, its rows can.
    if (n_passes > 1 && static_cast<uint64_t>(chunk) * pixels_of(p) / g->member.size() < (uint64_t{1} << 25))
        return group_render_bands(g, p, rgb8, progress, user, n_passes, chunk, first_pass);
    for (uint32_t pass = 0; pass < n_passes; ++pass) {
        const uint32_t k = pass + first_pass;  // sample_iteration, lib.rs:233-234
        const uint64_t s_begin = static_cast<uint64_t>(k) * chunk;
        if (s_begin + chunk > 0xFFFFFFFFull) return group_fail(g, RT_ERR_INVALID, "sample index overflow");
        size_t holder = 0;
        rc = group_sums(g, p, static_cast<uint32_t>(s_begin), chunk, false, &holder);
        if (rc != RT_OK) return rc;
        // mean, gamma, running average and quantisation (lib.rs:283-295) on the member whose GPU
        // holds the finished sums, straight from its memory; the image is readable after every pass
        rc = rt_resolve(g->member[holder], p, g->d_sums[holder], 1, chunk, running, k, rgb8);
        if (rc != RT_OK) return member_fail(g, holder, rc);
        if (progress) progress(first_pass + (pass + 1) * chunk, user);  // lib.rs:238-246
    }
    return RT_OK;
} RT_ABI_END_CTX(g)

int rt_group_get_stats(RtGroup* g, RtStats* out) try {
    if (!g || !out) return RT_ERR_INVALID;
    RtStats total{};
    for (size_t k = 0; k < g->member.size(); ++k) {
        RtStats s{};
        const int rc = rt_get_stats(g->member[k], &s);
        if (rc != RT_OK) return member_fail(g, k, rc);
        total.paths += s.paths;
        total.segments += s.segments;
        total.hits += s.hits;
        total.branch_visits += s.branch_visits;
        for (int f = 0; f < 3; ++f) total.prim_tests[f] += s.prim_tests[f];
        total.trace_launches += s.trace_launches;
        total.trace_ms += s.trace_ms;
        total.render_ms = std::max(total.render_ms, s.render_ms);  // members work side by side
    }
    *out = total;
    return RT_OK;
} RT_ABI_END_CTX(g)

int rt_group_reset_stats(RtGroup* g) try {
    if (!g) return RT_ERR_INVALID;
    for (RtContext* ctx : g->member) (void)rt_reset_stats(ctx);
    return RT_OK;
} RT_ABI_END_CTX(g)

}  // extern "C"
