// The exchange step of the sharded MD path as ONE kernel of direct peer writes ("tuned variant").
//
// Reference: the fragment batch is cut into contiguous per-device ranges (Calculators/device_strategy.py:84-127), one
// worker per device evaluates its range (Calculators/bonded.py:65-83, visnet_calculator.py:78-118 pickles the results
// back over a socket) and the host concatenates them (bonded.py:80-89).  Here every rank is a process with its own GPU
// and the "concatenate" is an all-gather of each rank's slot - forces of its rows + energies of its fragments, 2-3 KB.
//
// The library's default exchange is `torch.distributed.all_gather_into_tensor` (RCCL): a general-purpose collective
// behind a host enqueue and a stream hand-off.  At this size the step is pure latency, and xGMI is a point-to-point
// mesh (7 links per GPU): every rank can simply STORE its slot into each peer's gather buffer.  One launch per step:
//
//   workgroup q of rank r:   copy  send[0 .. slot)  ->  peer q's  data[par][r][0 .. slot)      (remote stores, xGMI)
//                            system-scope fence;  peer q's  flags[r] = step                    (release store)
//                            wait until  MY  flags[q] >= step                                   (acquire loads)
//
// so when the kernel retires every slot of this step has landed in THIS rank's buffer and the combine that follows on
// the same stream reads it after a kernel boundary.  Buffers are mapped into the peers with hipIpcGetMemHandle /
// hipIpcOpenMemHandle (one process per GPU); gather buffer and flags live in fine-grained memory so that a running
// kernel sees the peers' flag stores and the combine never reads a stale cached line of a slot a peer has rewritten.
// `par` = step & 1: a peer that runs ahead writes step s + 1 into the OTHER half; it can only reach step s + 2 after this rank's flag for s + 1, which this rank stores after its combine of step s (stream
// order) - no slot is overwritten while it may still be read.  The step counter is monotonic, so a flag never has to
// be reset.  A wait that lasts longer than `timeout` (default 5 s: a peer died) gives up and raises the handle's
// status instead of hanging the GPU.
//
// Bitwise the same gathered buffer as the RCCL / gloo path (a copy is a copy); validated with 2 / 4 / 8 real ranks
// sharing one GPU (tests/test_gpu_multirank.py) - IPC mappings work between processes on one device; the xGMI wire
// itself has not been driven (1-GPU boxes).
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "../../include/vsn.h"

namespace {

struct P2PView {
  int rank, world, slot;       // slot: floats per rank
  const float* send;           // [slot]  this rank's slot (its kernels write here)
  float* const* peer_data;     // [world] -> data[2][world][slot] of every rank (own entry = local pointer)
  unsigned* const* peer_flags; // [world] -> flags[world] of every rank
  const unsigned* my_flags;    // [world] fine-grained
  int* err;                    // set to the step whose wait timed out
  unsigned long long timeout_ticks;  // of the 100 MHz real-time counter
};

__global__ __launch_bounds__(256) void k_p2p_allgather(P2PView v, unsigned step) {
  const int q = blockIdx.x, tid = threadIdx.x;
  float* dst = v.peer_data[q] + ((size_t)(step & 1u) * v.world + v.rank) * v.slot;
  for (int i = tid; i < v.slot; i += blockDim.x) dst[i] = v.send[i];
  __threadfence_system();
  __syncthreads();
  if (tid == 0) {
    __hip_atomic_store(v.peer_flags[q] + v.rank, step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    // (unsigned difference: correct across the counter's wrap at 2^32 steps)
    while ((int)(__hip_atomic_load(v.my_flags + q, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - step) < 0) {
      __builtin_amdgcn_s_sleep(8);
      if (__builtin_amdgcn_s_memrealtime() - t0 > v.timeout_ticks) {
        *v.err = (int)step;
        break;
      }
    }
  }
}

}  // namespace

struct vsn_p2p {
  int device = 0, rank = 0, world = 1, slot = 0;
  unsigned step = 0;
  bool connected = false;
  float* data = nullptr;        // [2][world][slot]
  unsigned* flags = nullptr;    // [world], fine-grained
  float* send = nullptr;        // [slot]
  int* err = nullptr;           // device word
  float** d_peer_data = nullptr;
  unsigned** d_peer_flags = nullptr;
  std::vector<void*> opened;    // mappings of the peers' buffers (closed on destroy)
  double timeout_s = 5.0;
};

extern "C" int vsn_p2p_create(vsn_p2p_handle* out, int device_id, int rank, int world, int64_t slot_floats) {
  if (!out || world < 1 || rank < 0 || rank >= world || slot_floats <= 0 || slot_floats > (1 << 28)) return -22;
  if (hipSetDevice(device_id) != hipSuccess) return -19;
  vsn_p2p* p = new vsn_p2p();
  p->device = device_id;
  p->rank = rank;
  p->world = world;
  p->slot = (int)slot_floats;
  const size_t nd = (size_t)2 * world * slot_floats * sizeof(float);
  // gather buffer AND flags in fine-grained device memory: a peer's stores arrive over the fabric, not through this
  // GPU's L2s - fine-grained lines are never served stale from them (a few KB per step: no cost worth measuring)
  bool ok = hipExtMallocWithFlags((void**)&p->data, nd, hipDeviceMallocFinegrained) == hipSuccess &&
            hipExtMallocWithFlags((void**)&p->flags, (size_t)world * sizeof(unsigned), hipDeviceMallocFinegrained) ==
                hipSuccess &&
            hipMalloc((void**)&p->send, (size_t)slot_floats * sizeof(float)) == hipSuccess &&
            hipMalloc((void**)&p->err, sizeof(int)) == hipSuccess &&
            hipMalloc((void**)&p->d_peer_data, (size_t)world * sizeof(float*)) == hipSuccess &&
            hipMalloc((void**)&p->d_peer_flags, (size_t)world * sizeof(unsigned*)) == hipSuccess;
  if (!ok) {
    (void)hipGetLastError();
    vsn_p2p_destroy(p);
    return -12;
  }
  hipMemset(p->data, 0, nd);
  hipMemset(p->flags, 0, (size_t)world * sizeof(unsigned));
  hipMemset(p->send, 0, (size_t)slot_floats * sizeof(float));
  hipMemset(p->err, 0, sizeof(int));
  hipDeviceSynchronize();
  *out = p;
  return 0;
}

extern "C" int vsn_p2p_export(vsn_p2p_handle p, void* handle_bytes) {
  if (!p || !handle_bytes) return -22;
  static_assert(2 * sizeof(hipIpcMemHandle_t) <= VSN_P2P_HANDLE_BYTES, "handle record too small");
  if (hipSetDevice(p->device) != hipSuccess) return -19;
  hipIpcMemHandle_t h[2];
  if (hipIpcGetMemHandle(&h[0], p->data) != hipSuccess || hipIpcGetMemHandle(&h[1], p->flags) != hipSuccess) {
    (void)hipGetLastError();
    return -5;
  }
  memset(handle_bytes, 0, VSN_P2P_HANDLE_BYTES);
  memcpy(handle_bytes, h, sizeof(h));
  return 0;
}

extern "C" int vsn_p2p_connect(vsn_p2p_handle p, const void* all_handles) {
  if (!p || (p->world > 1 && !all_handles)) return -22;
  if (p->connected) return -16;
  if (hipSetDevice(p->device) != hipSuccess) return -19;
  std::vector<float*> pd((size_t)p->world, nullptr);
  std::vector<unsigned*> pf((size_t)p->world, nullptr);
  for (int q = 0; q < p->world; ++q) {
    if (q == p->rank) {  // (a process cannot open its own handle)
      pd[q] = p->data;
      pf[q] = p->flags;
      continue;
    }
    hipIpcMemHandle_t h[2];
    memcpy(h, (const char*)all_handles + (size_t)q * VSN_P2P_HANDLE_BYTES, sizeof(h));
    void *a = nullptr, *b = nullptr;
    if (hipIpcOpenMemHandle(&a, h[0], hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
      (void)hipGetLastError();
      return -5;
    }
    p->opened.push_back(a);
    if (hipIpcOpenMemHandle(&b, h[1], hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
      (void)hipGetLastError();
      return -5;
    }
    p->opened.push_back(b);
    pd[q] = (float*)a;
    pf[q] = (unsigned*)b;
  }
  if (hipMemcpy(p->d_peer_data, pd.data(), pd.size() * sizeof(float*), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(p->d_peer_flags, pf.data(), pf.size() * sizeof(unsigned*), hipMemcpyHostToDevice) != hipSuccess)
    return -5;
  p->connected = true;
  return 0;
}

extern "C" float* vsn_p2p_send_buffer(vsn_p2p_handle p) { return p ? p->send : nullptr; }

extern "C" float* vsn_p2p_gather_buffer(vsn_p2p_handle p, int parity) {
  return p ? p->data + (size_t)(parity & 1) * p->world * p->slot : nullptr;
}

extern "C" int vsn_p2p_set_timeout(vsn_p2p_handle p, double seconds) {
  if (!p || !(seconds > 0.0)) return -22;
  p->timeout_s = seconds;
  return 0;
}

extern "C" int vsn_p2p_allgather(vsn_p2p_handle p, void* stream, float** dev_gathered_out) {
  if (!p || !p->connected) return -22;
  if (hipSetDevice(p->device) != hipSuccess) return -19;
  p->step += 1;
  if (p->step == 0) p->step = 2;  // (wrap: keep the parity sequence, never hand out the initial flag value 0)
  P2PView v;
  v.rank = p->rank;
  v.world = p->world;
  v.slot = p->slot;
  v.send = p->send;
  v.peer_data = p->d_peer_data;
  v.peer_flags = p->d_peer_flags;
  v.my_flags = p->flags;
  v.err = p->err;
  v.timeout_ticks = (unsigned long long)(p->timeout_s * 100e6);
  hipLaunchKernelGGL(k_p2p_allgather, dim3(p->world), dim3(256), 0, (hipStream_t)stream, v, p->step);
  if (dev_gathered_out) *dev_gathered_out = vsn_p2p_gather_buffer(p, (int)(p->step & 1u));
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

extern "C" int vsn_p2p_status(vsn_p2p_handle p, void* stream) {
  if (!p) return -22;
  if (hipSetDevice(p->device) != hipSuccess) return -19;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -5;
  int e = 0;
  if (hipMemcpy(&e, p->err, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -5;
  return e;  // 0 = every wait completed; else the (first reported) step whose wait timed out
}

extern "C" void vsn_p2p_destroy(vsn_p2p_handle p) {
  if (!p) return;
  hipSetDevice(p->device);
  hipDeviceSynchronize();
  for (void* m : p->opened) hipIpcCloseMemHandle(m);
  hipFree(p->data);
  hipFree(p->flags);
  hipFree(p->send);
  hipFree(p->err);
  hipFree(p->d_peer_data);
  hipFree(p->d_peer_flags);
  delete p;
}
