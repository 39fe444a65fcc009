// Static device workspace of the allocation-free entry point (scratch.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

// 384 MiB per GPU: one 10 s utterance (two 64-frame units) of the full-size decoder in fp32 (185 MiB per unit: three
// ping-pong tensors plus the two activated copies the Winograd kernels read) and bf16, four units in bf16x3; many units of
// smaller topologies.
#define HIPVOC_SCRATCH_BYTES (384u << 20)
#define HIPVOC_SCRATCH_MAX_DEVICES 64

namespace hipvoc {
// Device address and size of `device`'s copy of the scratch array.  The first call for a device resolves the symbol and
// must be made with that device current (hipvoc_load does); later calls return the cached pointer.
hipError_t scratch_bind(int device, void **ptr, size_t *bytes);
// Brackets of one pass over `device`'s array, enqueued on `stream`: begin makes the stream wait for the previous pass on
// that device (whatever handle / stream it ran on) and takes the device's host lock; end records the pass's completion
// and releases the lock.  Neither allocates (the event is created by scratch_bind).  end must follow a successful begin.
hipError_t scratch_pass_begin(int device, hipStream_t stream);
hipError_t scratch_pass_end(int device, hipStream_t stream);
}
