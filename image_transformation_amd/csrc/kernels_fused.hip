// The LANCZOS path as ONE path, launch level (round 4): a FUSED launch = the marching resample of the NEXT chunk of
// canvases and the composite of the PREVIOUS one, as two ROLES of one kernel.
//
// Why.  The reference resamples and blends an object back to back (compositor.py:20-21).  Here the resample kernel
// is bound by instruction issue (84 % of vector issue in its loop) and keeps the memory system at a third of its
// rate, the composite kernel is bound by memory and leaves the vector units idle; run one after the other they add
// up (C3 placements, 16 canvases per call: 41.9 + 18.9 us per canvas).  Overlapping them through streams and events
// was measured and lost (profiles/r04_pipeline_streams.txt: a cross-stream edge costs 15-50 us on this runtime).
// Inside ONE launch nothing has to be ordered: the two roles touch disjoint memory -- the composite role reads layers
// that an EARLIER launch resampled (the kernel boundary is the release / acquire), the resample role writes the layers
// the NEXT launch composites.  A call over chunks c0, c1, ... of its canvases becomes
//     resample(c0) | fused{resample(c1), composite(c0)} | fused{resample(c2), composite(c1)} | ... | composite(c_last)
// on one stream, and the memory-bound pages run in the wave slots and memory cycles the issue-bound units leave.
//
// Mapping.  One workgroup = 256 threads either way: a resample workgroup is one work unit of the marching kernel
// (kernels_resample.hip), a composite workgroup is FOUR consecutive 4 KiB pages of one canvas, one per wave
// (kernels_composite.hip's one-wave-per-page body, unchanged).  The roles are interleaved over blockIdx.x in
// proportion (workgroup b is a composite one iff floor((b + 1) nC / T) > floor(b nC / T)), so that both kinds are
// resident on every CU for the whole launch -- dispatched one after the other the composite pages would only start
// when the last resample unit has been handed out.  Both bodies are the standalone kernels' own text
// (composite_body.inc, resample_march_body.inc): the fused launch computes the same bytes by construction.
// Register / LDS budget = the resample role's (94 VGPRs, the call's dynamic LDS): 5 workgroups per CU; a composite
// workgroup holds such a slot for ~1 us, a resample unit for ~10.
#include <atomic>
#include <cstdlib>

#include "composite_device.h"
#include "resample_device.h"

namespace mic {

struct FusedArgs {
    const RsMarch *rs_jobs;   // resample role: table entries [0, n_rs) of kRsUnitsPerEntry workgroups each
    int32_t n_rs;
    int32_t n_jobs;           // composite role: jobs [0, n_jobs) of the (class-sorted) device job table ...
    const Job *jobs;
    const Layer *layers;
    int32_t wg_per_job;       // ... each cut into this many workgroups of 4 pages (pitch / 4)
    int32_t pad;
};

template <bool ALIGNED>
__global__ __launch_bounds__(256, MIC_RS_WAVES) void fused_kernel(const FusedArgs A) {
    constexpr bool SOLID = true;  // (the fused path takes canvases over an opaque solid background: fill_solid's)
    const uint32_t b = blockIdx.x;
    const uint32_t n_r = (uint32_t)A.n_rs * (uint32_t)kRsUnitsPerEntry, n_c = (uint32_t)A.n_jobs * (uint32_t)A.wg_per_job;
    const uint32_t total = n_r + n_c;
    uint32_t c_before, c_after;
    if (A.pad == 0) {  // EXPERIMENT knob (MIC_FUSE_MODE): 0 fine interleave, 1 groups of 8 (XCD-preserving), 2 R then C, 3 C then R
        c_before = (uint32_t)((uint64_t)b * n_c / total); c_after = (uint32_t)((uint64_t)(b + 1) * n_c / total);
    } else if (A.pad == 1) {
        const uint32_t g = b >> 3, cg = (n_c + 7) >> 3, ng = ((n_r + 7) >> 3) + cg;  // groups of 8 workgroups, one role per group
        const uint32_t gb = (uint32_t)((uint64_t)g * cg / ng), ga = (uint32_t)((uint64_t)(g + 1) * cg / ng);
        if (ga > gb) { c_before = gb * 8 + (b & 7); c_after = c_before + 1; if (c_before >= n_c) return; }
        else { c_before = b - ((g - gb) * 8 + (b & 7)); c_after = c_before; if (b - c_before >= n_r) return; }
    } else if (A.pad == 2) {
        c_before = b < n_r ? 0 : b - n_r; c_after = b < n_r ? 0 : c_before + 1;
    } else {
        c_before = b < n_c ? b : n_c; c_after = b < n_c ? b + 1 : n_c;
    }
    const int tid = threadIdx.x;
    if (c_after > c_before) {
        // ---- composite role: workgroup c_before of the composite part
        const uint32_t ji = c_before / (uint32_t)A.wg_per_job, p4 = c_before - ji * (uint32_t)A.wg_per_job;
        const Job job = A.jobs[ji];
        // (the wave index as a scalar: everything the body derives from the page -- rows, columns, culling -- is
        // wave-uniform and must live in scalar registers, as in the one-wave workgroups of the standalone kernel)
        const int page = (int)(p4 * 4u) + __builtin_amdgcn_readfirstlane(tid >> 6);
        if (page >= job.n_pages) return;  // (per wave: the composite body has no workgroup barrier)
        const int lane = tid & 63;
        const Layer *jl = A.layers + job.layer_begin;
#include "composite_body.inc"
    } else {
        // ---- resample role: workgroup (b - c_before) of the resample part
        const uint32_t r = b - c_before;
        const RsMarch J = A.rs_jobs[r / (uint32_t)kRsUnitsPerEntry];
        const int bx = (int)(r % (uint32_t)kRsUnitsPerEntry);
#include "resample_march_body.inc"
    }
}

// rs_jobs_dev / n_rs: the marching entries of the NEXT chunk; jobs_dev / n_jobs: the previous chunk's slice of the job
// table (every job over an opaque solid background; all_aligned: every one of them W % 4 == 0 on a 16-byte aligned
// canvas); pitch: pages per job rounded up to 8, as launch_composite takes it.
hipError_t launch_fused(const RsMarch *rs_jobs_dev, int n_rs, size_t lds_bytes, const Job *jobs_dev, const Layer *layers_dev,
                        int n_jobs, int pitch, bool all_aligned, hipStream_t stream) {
    static const int mode = [] { const char *e = getenv("MIC_FUSE_MODE"); return e ? atoi(e) : 0; }();
    if (n_rs <= 0 || n_jobs <= 0 || pitch <= 0) return hipErrorInvalidValue;
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(fused_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)kRsMarchMaxLds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(fused_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)kRsMarchMaxLds);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set[dev].store(true, std::memory_order_release);
    }
    FusedArgs A{};
    A.rs_jobs = rs_jobs_dev; A.n_rs = n_rs;
    A.jobs = jobs_dev; A.layers = layers_dev; A.n_jobs = n_jobs; A.wg_per_job = pitch / 4;
    A.pad = mode;
    uint64_t total = (uint64_t)n_rs * kRsUnitsPerEntry + (uint64_t)n_jobs * (uint64_t)A.wg_per_job;
    if (mode == 1) total = ((((uint64_t)n_rs * kRsUnitsPerEntry + 7) >> 3) + (((uint64_t)n_jobs * (uint64_t)A.wg_per_job + 7) >> 3)) * 8;
    if (total > 0x7fffffffull) return hipErrorInvalidValue;
    if (all_aligned) hipLaunchKernelGGL(fused_kernel<true>, dim3((unsigned)total), dim3(256), lds_bytes, stream, A);
    else hipLaunchKernelGGL(fused_kernel<false>, dim3((unsigned)total), dim3(256), lds_bytes, stream, A);
    return hipGetLastError();
}

}  // namespace mic
