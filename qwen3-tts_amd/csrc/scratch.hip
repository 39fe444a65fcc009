// scratch.hip -- the statically allocated workspace of hipvoc_decode_units_device_noalloc (include/hipvoc.h).
//
// Callers that cannot call hipMalloc (code under HIP graph capture, tight serving loops, harnesses that forbid
// allocation) get one entry point whose activation workspace -- everything the decode kernels write besides the caller's
// output: dequantised latents, transformer activations, the two ping-pong tensors of the decoder stack -- lives in the
// file-scope device array below instead of hipMalloc'ed buffers.  Every GPU the library runs on holds its own copy of the
// array, at its own address, until the process ends.
//
// The array's NAME is a host-side stand-in, not a device address: host code never passes it to a kernel or a memset.
// scratch_bind() resolves the device address once per device with hipGetSymbolAddress (that device current), and the
// library uses that pointer everywhere.
#include <hip/hip_runtime.h>

#include <mutex>

#include "scratch.h"

static __device__ __align__(256) unsigned char scratch[HIPVOC_SCRATCH_BYTES];

namespace hipvoc {

static std::mutex g_scratch_mu;
static void *g_scratch_ptr[HIPVOC_SCRATCH_MAX_DEVICES];      // per device: resolved address of that device's copy
// Ordering of the passes that share one device's array.  Every handle enqueues on its own non-blocking stream, so a host
// that serialises its CALLS (as the reference's per-worker mutex does) would still have two handles' kernels overlapping
// on the GPU.  One disable-timing event per device, created at bind time (hipvoc_load), chains them: a pass waits for the
// event of the previous pass on that device and records it again behind its own last kernel.  The per-device mutex covers
// wait .. record, so the chain has one order.
static hipEvent_t g_scratch_ev[HIPVOC_SCRATCH_MAX_DEVICES];
static std::mutex g_scratch_order_mu[HIPVOC_SCRATCH_MAX_DEVICES];

hipError_t scratch_bind(int device, void **ptr, size_t *bytes) {
    *ptr = nullptr;
    *bytes = 0;
    if (device < 0 || device >= HIPVOC_SCRATCH_MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        int cur = -1;
        hipError_t e = hipGetDevice(&cur);
        if (e != hipSuccess) return e;
        if (cur != device) return hipErrorInvalidDevice;      // the caller makes the device current first
        void *p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        hipEvent_t ev = nullptr;
        e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e != hipSuccess) return e;
        g_scratch_ev[device] = ev;
        g_scratch_ptr[device] = p;
    }
    *ptr = g_scratch_ptr[device];
    *bytes = HIPVOC_SCRATCH_BYTES;
    return hipSuccess;
}

hipError_t scratch_pass_begin(int device, hipStream_t stream) {
    if (device < 0 || device >= HIPVOC_SCRATCH_MAX_DEVICES || !g_scratch_ev[device]) return hipErrorInvalidDevice;
    g_scratch_order_mu[device].lock();
    hipError_t e = hipStreamWaitEvent(stream, g_scratch_ev[device], 0);      // (an event never recorded yet: no wait)
    if (e != hipSuccess) g_scratch_order_mu[device].unlock();
    return e;
}

hipError_t scratch_pass_end(int device, hipStream_t stream) {
    hipError_t e = hipEventRecord(g_scratch_ev[device], stream);
    g_scratch_order_mu[device].unlock();
    return e;
}

}  // namespace hipvoc
