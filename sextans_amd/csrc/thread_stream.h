// thread_stream.h -- synchronous copies and waits without the LEGACY default stream / the whole device.
//
// hipMemcpy / hipMemset wait on the process-wide legacy stream, and HIP fails them -- in EVERY host thread -- while ANY stream of the
// device is being captured into a hipGraph ("operation would make the legacy stream depend on a capturing blocking stream"), and
// invalidates that capture on top.  The host-buffer entry points capture their rp_time loop (engine.hip: run_repeats), the one-time
// plan builders copy small tables back and forth: two engines used by two host threads (re-entrant per handle; the
// thread-per-GPU model of examples/dist_spmm.cpp) broke each other.  Found by tests/test_concurrency_gpu.py in round 5;
// tools/capture_race.py reproduces it.  Inside the library a "synchronous" copy is therefore an asynchronous copy on the calling
// thread's own default stream followed by a wait for that stream; the builders' kernels (launched on stream 0) run on the same
// stream because the library is compiled with -fgpu-default-stream=per-thread (sextans_amd/build.py), so the order is kept.
// Include AFTER <hip/hip_runtime.h> and any hipcub header.
#pragma once
#include <hip/hip_runtime.h>

namespace sx {
inline hipError_t memcpy_on_thread_stream(void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
    if (bytes == 0) return hipSuccess;
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, hipStreamPerThread);
    return e != hipSuccess ? e : hipStreamSynchronize(hipStreamPerThread);
}
inline hipError_t memset_on_thread_stream(void *dst, int value, size_t bytes) {
    if (bytes == 0) return hipSuccess;
    const hipError_t e = hipMemsetAsync(dst, value, bytes, hipStreamPerThread);
    return e != hipSuccess ? e : hipStreamSynchronize(hipStreamPerThread);
}
}  // namespace sx

#undef hipMemcpy   // (-fgpu-default-stream=per-thread maps them to the _spt entry points, which wait on the legacy stream all the same)
#undef hipMemset
#define hipMemcpy(dst, src, bytes, kind) sx::memcpy_on_thread_stream((dst), (src), (bytes), (kind))
#define hipMemset(dst, value, bytes) sx::memset_on_thread_stream((dst), (value), (bytes))
// hipDeviceSynchronize is refused likewise ("operation not permitted when stream is capturing") while another host thread captures.  What
// the library's builders wait for is their OWN work -- kernels and copies on stream 0 = the calling thread's default stream -- so the
// wait is for that stream.  (Write (hipDeviceSynchronize)() where the whole device is meant.)
#define hipDeviceSynchronize() hipStreamSynchronize(hipStreamPerThread)
