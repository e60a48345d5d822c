// Streams that really run side by side.
//
// HIP multiplexes every stream of a process onto a handful of hardware queues (4 by default); which queue a new stream lands on
// depends on what else the process created and destroyed before.  Two streams that share a queue execute their kernels one after the
// other, whatever the events between them say.  Measured (tools/order_probe.py, profiles/r02_stream_queues.txt): the same two-lane
// vgg_heads_m b32 forward takes 5.1 ms when its lanes sit on different queues and 7.1 ms when they share one -- the 2nd, 3rd, ...
// engine of a process flipped between the two depending on the creation history, and GPU_MAX_HW_QUEUES only moved the pattern.
//
// So the library does not trust a fresh hipStreamCreate: vgh_stream_acquire hands out a stream that was MEASURED to overlap with
// every stream in `avoid` (two 150 us spin kernels launched back to back on the pair: side by side they take ~150 us together, on one
// queue ~300 us).  Candidates that fail stay parked (alive, so the runtime's queue assignment keeps moving on) and are offered to later
// requests; released streams go back to the same per-device park instead of being destroyed.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <mutex>
#include <vector>

#include "vgh_internal.h"

namespace {

__global__ void spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();  // constant 100 MHz counter
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}

// many more workgroups than the chip holds at once (80 KiB of LDS each: two per CU), each spinning briefly: two such kernels on
// streams whose workgroups the dispatcher can interleave finish together; when the second kernel's workgroups only start once the
// first kernel's have all been dispatched, the first one finishes in half the time of the pair
__global__ void spin_grid_kernel(long long ticks) {
    extern __shared__ float pad_lds[];
    if (threadIdx.x == 0) pad_lds[0] = 0.0f;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}

constexpr long long kSpinTicks = 15000;  // 150 us
constexpr int kMaxDev = 16, kMaxTries = 10;
// Upper bound on the streams one (device, priority) park ever holds.  With H hardware queues at most H - n_avoid candidates can be
// "perfect"; when none is (e.g. main + 3 lanes already occupy the 4 default queues) every further hipStreamCreate lands on one of the
// same H queues, so a full park already contains a member of every queue class and creating more only leaks streams and lengthens the
// probe of the next acquire.
constexpr int kMaxPark = 16;

struct DeviceGuard {  // restores the caller's current device on every exit path
    int prev = -1;
    bool switched = false;
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};

std::mutex g_mu;
std::vector<hipStream_t> g_park[2][kMaxDev];  // [normal | lowest priority]

// true when kernels queued on a and b at the same time overlap
bool runs_concurrently(hipStream_t a, hipStream_t b) {
    if (a == b) return false;
    // untimed pass: pays the one-off launch set-up of a new stream and leaves both idle
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, a, 100);
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, b, 100);
    (void)hipStreamSynchronize(a);
    (void)hipStreamSynchronize(b);
    // a pair on one queue can never look concurrent (>= 2 x 150 us); a concurrent pair can look serial when the host is preempted
    // between the two launches -- so "serial" is only believed when seen twice
    for (int attempt = 0; attempt < 2; ++attempt) {
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, a, kSpinTicks);
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, b, kSpinTicks);
        (void)hipStreamSynchronize(a);
        (void)hipStreamSynchronize(b);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        (void)hipGetLastError();
        if (us < 1.6 * (double)kSpinTicks / 100.0) return true;
    }
    return false;
}

// Head-of-line blocking inside a compute pipe (r06: the "starved side stream" of r05, profiles/r06_starved_side_stream_classes.txt, r06_queue_trace.txt).
// The runtime gives every priority class its own hardware queues -- a low-priority stream never SHARES a queue with a normal one, which is all runs_concurrently can
// see -- but the queues are spread over the device's compute pipes in creation order (the trace shows normal queues 1 - 4, low-priority queues 5, 6, ...), so the first
// low-priority queue sits on the pipe of the first normal queue.  A pipe places the workgroups of ONE dispatch at a time: while a lane keeps that pipe busy with
// kernels whose workgroups do not all fit on the chip (every conv of the network), a kernel of the low-priority queue behind it is not even looked at, whatever free
// wave slots the chip has -- the post-network stages then only run when the next forward stalls at its prediction guard (14.0 instead of 11.5 ms per forward).  On any
// OTHER pipe the same low-priority kernels slip in beside the network.  Measured here the way it bites: `a` gets two multi-round kernels whose workgroups are limited
// by LDS (wave slots stay free), `b` one single-wave kernel that becomes eligible when the first of them starts; on another pipe it is done in ~10 us, behind `a`'s
// pipe only when `a`'s dispatches have been placed.
bool blocked_behind(hipStream_t a, hipStream_t b) {
    if (a == b) return true;
    static bool attr = false;
    const int lds = 80 * 1024;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)spin_grid_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr = true;
    }
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    hipEvent_t e0, ea, eb;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&ea) != hipSuccess || hipEventCreate(&eb) != hipSuccess) return false;
    const int grid = cus * 2 * 6;  // six rounds of the LDS-limited residency, 10 us each
    int votes = 0;
    for (int pass = 0; pass < 3; ++pass) {  // pass 0 warms both streams up; "blocked" is believed when seen in both timed passes
        (void)hipStreamSynchronize(a);
        (void)hipStreamSynchronize(b);
        (void)hipEventRecord(e0, a);
        hipLaunchKernelGGL(spin_grid_kernel, dim3(grid), dim3(64), lds, a, 1000);
        hipLaunchKernelGGL(spin_grid_kernel, dim3(grid), dim3(64), lds, a, 1000);
        (void)hipEventRecord(ea, a);
        (void)hipStreamWaitEvent(b, e0, 0);
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, b, 100);
        (void)hipEventRecord(eb, b);
        (void)hipStreamSynchronize(a);
        (void)hipStreamSynchronize(b);
        float ta = 0, tb = 0;
        (void)hipEventElapsedTime(&ta, e0, ea);
        (void)hipEventElapsedTime(&tb, e0, eb);
        if (pass > 0 && tb > 0.3f * ta) ++votes;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(ea);
    (void)hipEventDestroy(eb);
    (void)hipGetLastError();
    return votes == 2;
}

int overlap_score(hipStream_t c, const hipStream_t* avoid, int n_avoid, bool low_priority) {
    int score = 0;
    for (int i = 0; i < n_avoid; ++i)
        // (the pipe test covers the first two entries -- the caller's stream and the first lane, what a two-lane forward runs on: with three or four lanes in use
        //  every pipe carries one and no low-priority queue can be clear of them all)
        if (runs_concurrently(avoid[i], c) && !(low_priority && i < 2 && blocked_behind(avoid[i], c))) score += 1 << (n_avoid - 1 - i);  // earlier entries of `avoid` weigh more
    return score;
}

}  // namespace

int vgh_stream_acquire_internal(int device, const hipStream_t* avoid, int n_avoid, hipStream_t* out, bool low_priority) {
    VGH_REQUIRE(out && device >= 0 && device < kMaxDev && n_avoid >= 0 && n_avoid <= 8, "stream_acquire: bad argument");
    std::lock_guard<std::mutex> lk(g_mu);
    DeviceGuard guard;
    VGH_HIP(hipGetDevice(&guard.prev));
    if (guard.prev != device) {
        VGH_HIP(hipSetDevice(device));
        guard.switched = true;
    }
    const int perfect = (1 << n_avoid) - 1;
    std::vector<hipStream_t>& park = g_park[low_priority ? 1 : 0][device];
    int least = 0, greatest = 0;
    if (low_priority) (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    int best = -1, best_score = -1;
    for (int i = 0; i < (int)park.size() && best_score < perfect; ++i) {
        const int s = overlap_score(park[i], avoid, n_avoid, low_priority);
        if (s > best_score) best = i, best_score = s;
    }
    for (int t = 0; t < kMaxTries && best_score < perfect && (int)park.size() < kMaxPark; ++t) {
        hipStream_t c;
        if (low_priority)
            VGH_HIP(hipStreamCreateWithPriority(&c, hipStreamNonBlocking, least));
        else
            VGH_HIP(hipStreamCreateWithFlags(&c, hipStreamNonBlocking));
        park.push_back(c);
        const int s = overlap_score(c, avoid, n_avoid, low_priority);
        if (s > best_score) best = (int)park.size() - 1, best_score = s;
    }
    VGH_REQUIRE(best >= 0, "stream_acquire: no candidate stream");
    if (best_score < perfect)  // not silent: a lane or side stream that shares a hardware queue with the stream it should run beside costs 10 - 25 % of a forward
        fprintf(stderr, "[vgh] stream_acquire (%s priority): no candidate overlaps with all %d streams to avoid (best mask 0x%x of 0x%x, %zu parked candidates)\n",
                low_priority ? "low" : "normal", n_avoid, best_score, perfect, park.size());
    *out = park[best];
    park.erase(park.begin() + best);
    return VGH_OK;
}

void vgh_stream_release_internal(int device, hipStream_t s, bool low_priority) {
    if (!s || device < 0 || device >= kMaxDev) return;
    std::lock_guard<std::mutex> lk(g_mu);
    g_park[low_priority ? 1 : 0][device].push_back(s);
}

extern "C" {

int vgh_stream_acquire(int device, void* const* avoid, int n_avoid, void** stream_out) {
    VGH_REQUIRE(stream_out && (avoid || n_avoid == 0), "stream_acquire: null argument");
    hipStream_t s = nullptr;
    const int rc = vgh_stream_acquire_internal(device, (const hipStream_t*)avoid, n_avoid, &s, false);
    *stream_out = (void*)s;
    return rc;
}

int vgh_stream_release(int device, void* stream) {
    vgh_stream_release_internal(device, (hipStream_t)stream, false);
    return VGH_OK;
}

int vgh_streams_overlap(void* a, void* b) { return runs_concurrently((hipStream_t)a, (hipStream_t)b) ? 1 : 0; }

// 1 when a kernel on `b` cannot start while `a`'s dispatches are being placed (same compute pipe: see blocked_behind)
int vgh_stream_blocked_behind(void* a, void* b) { return blocked_behind((hipStream_t)a, (hipStream_t)b) ? 1 : 0; }

#ifdef VGH_EXPERIMENTS
// first-kernel completion time / pair completion time for two multi-round kernels launched back to back on a and b:
// ~1.0 = their workgroups interleave, ~0.5 = b's only start when a's are all dispatched.  *1000 (integer per-mille).
int vgh_streams_interleave_permille(void* a, void* b) {
    static bool attr = false;
    const int lds = 80 * 1024;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)spin_grid_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr = true;
    }
    hipStream_t sa = (hipStream_t)a, sb = (hipStream_t)b;
    hipEvent_t e0, ea, eb;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&ea) != hipSuccess || hipEventCreate(&eb) != hipSuccess) return -1;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int grid = cus * 2 * 4;  // 4 rounds of the resident capacity
    int result = -1;
    for (int pass = 0; pass < 2; ++pass) {  // first pass warms both streams up
        (void)hipStreamSynchronize(sa);
        (void)hipStreamSynchronize(sb);
        (void)hipEventRecord(e0, sa);
        hipLaunchKernelGGL(spin_grid_kernel, dim3(grid), dim3(64), lds, sa, 2000);  // 20 us per workgroup
        (void)hipEventRecord(ea, sa);
        hipLaunchKernelGGL(spin_grid_kernel, dim3(grid), dim3(64), lds, sb, 2000);
        (void)hipEventRecord(eb, sb);
        (void)hipStreamSynchronize(sa);
        (void)hipStreamSynchronize(sb);
        float ta = 0, tb = 0, tab = 0;
        (void)hipEventElapsedTime(&ta, e0, ea);
        (void)hipEventElapsedTime(&tab, ea, eb);
        tb = ta + tab;
        result = tb > 0 ? (int)(1000.0f * ta / tb) : -1;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(ea);
    (void)hipEventDestroy(eb);
    (void)hipGetLastError();
    return result;
}

#endif

int vgh_stream_spin(void* stream, int microseconds) {
    VGH_REQUIRE(microseconds >= 0 && microseconds <= 100000, "stream_spin: 0..100000 us");
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long)microseconds * 100);
    VGH_HIP(hipGetLastError());
    return VGH_OK;
}

}  // extern "C"
