// Launch plans: record the device work of one train step once, re-issue it with ONE host call per step.
//
// The reference pays one host call per train step (`sess.run(train_op)`, luminoth/train.py:235-247): TensorFlow's
// executor walks a graph that was built once.  The Python host above this library walks its layer lists every step
// instead — ~250 launches, each a Python -> ctypes round trip of 20-25 us — and finishes enqueueing a 7 ms step only
// ~0.5 ms before the GPU finishes executing it (profiles/r03_bench_phases.json).  A plan is this library's counterpart of
// the built graph: while a thread records (lmh_plan_begin .. lmh_plan_end), every kernel launch, memset, event record
// and stream-to-stream wait that goes through lmh_launch_raw / lmh_memset_async / lmh_stream_wait_stream /
// lmh_event_record is executed as usual AND appended to the plan with a private copy of its argument values;
// lmh_plan_run re-issues the identical sequence — same kernels, same grids, same argument values, same streams, same
// cross-stream dependencies — straight from C (hipLaunchKernel: 2-3 us per launch).  Nothing is skipped or cached: a
// replayed step launches exactly the kernels the recorded step launched.
//
// It is NOT a HIP graph: hipGraphLaunch on ROCm 7.2 replayed the same step slower than the eager launches (DESIGN.md
// §4), serialising the captured streams.  A plan keeps the streams and their events as they are and only removes the
// interpreter from the launch path.
//
// Contract with the caller (luminoth_amd/plan.py): every device pointer recorded in a plan must stay valid and must mean
// the same thing at replay — the host keeps every tensor whose address entered a launch alive for as long as the plan
// lives and feeds per-step inputs through buffers at fixed addresses.  Scalars that change from step to step (learning
// rate) stay outside the recorded region.
#include <vector>

#include "lmh_common.h"

namespace {

enum NodeKind { NODE_KERNEL = 0, NODE_MEMSET = 1, NODE_EVENT_RECORD = 2, NODE_STREAM_WAIT = 3, NODE_MEMCPY = 4 };

struct PlanNode {
  int kind;
  hipStream_t st;
  // kernel
  const void* fn;
  dim3 grid, block;
  unsigned shmem;
  uint32_t arg_first;   // index into Plan::arg_off
  int nargs;
  // memset / device-to-device copy (ptr <- src)
  void* ptr;
  const void* src;
  int value;
  size_t bytes;
  // event record (on st) / stream wait (st waits for ev)
  hipEvent_t ev;
};

struct Plan {
  std::vector<PlanNode> nodes;
  std::vector<char> blob;            // argument values, each at its natural alignment
  std::vector<uint32_t> arg_off;     // offsets into blob
  std::vector<void*> arg_ptr;        // blob pointers (filled by finalize)
  std::vector<hipEvent_t> owned;     // events created for recorded stream-to-stream waits
  bool finalized = false;
  int failed = 0;
};

thread_local Plan* g_rec = nullptr;

}  // namespace

// ------------------------------------------------------------------------------------------ issue + record --------
void lmh_launch_raw(const void* fn, dim3 grid, dim3 block, unsigned shmem, hipStream_t st, void** args,
                    const size_t* sizes, const size_t* aligns, int nargs) {
  if (g_rec) {
    Plan* p = g_rec;
    PlanNode n{};
    n.kind = NODE_KERNEL;
    n.st = st;
    n.fn = fn;
    n.grid = grid;
    n.block = block;
    n.shmem = shmem;
    n.arg_first = (uint32_t)p->arg_off.size();
    n.nargs = nargs;
    for (int i = 0; i < nargs; ++i) {
      const size_t a = aligns[i] < 16 ? 16 : aligns[i];
      size_t off = (p->blob.size() + a - 1) / a * a;
      p->blob.resize(off + sizes[i]);
      memcpy(p->blob.data() + off, args[i], sizes[i]);
      p->arg_off.push_back((uint32_t)off);
    }
    p->nodes.push_back(n);
  }
  (void)hipLaunchKernel(fn, grid, block, args, shmem, st);
}

hipError_t lmh_memset_async(void* ptr, int value, size_t bytes, hipStream_t st) {
  if (g_rec) {
    PlanNode n{};
    n.kind = NODE_MEMSET;
    n.st = st;
    n.ptr = ptr;
    n.value = value;
    n.bytes = bytes;
    g_rec->nodes.push_back(n);
  }
  return hipMemsetAsync(ptr, value, bytes, st);
}

hipError_t lmh_memcpy_d2d_async(void* dst, const void* src, size_t bytes, hipStream_t st) {
  if (g_rec) {
    PlanNode n{};
    n.kind = NODE_MEMCPY;
    n.st = st;
    n.ptr = dst;
    n.src = src;
    n.bytes = bytes;
    g_rec->nodes.push_back(n);
  }
  return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st);
}

// `waiter` waits for everything enqueued on `signaler` so far (hipEventRecord + hipStreamWaitEvent).  Outside a
// recording the event comes from a small per-thread ring of timing-disabled events: a wait captures the event's state
// when it is enqueued, so re-recording an event 32 calls later does not disturb it.  While recording, the plan gets an
// event of its own per wait (re-recorded at every replay).  The host-side cost of the same thing through torch
// (Stream.wait_stream: a fresh Event object each time) is ~9 us; the train step does it once per trainable layer.
extern "C" int lmh_stream_wait_stream(lmh_stream_t waiter, lmh_stream_t signaler) {
  hipEvent_t e;
  if (g_rec) {
    LMH_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    g_rec->owned.push_back(e);
    PlanNode r{};
    r.kind = NODE_EVENT_RECORD;
    r.st = (hipStream_t)signaler;
    r.ev = e;
    g_rec->nodes.push_back(r);
    PlanNode w{};
    w.kind = NODE_STREAM_WAIT;
    w.st = (hipStream_t)waiter;
    w.ev = e;
    g_rec->nodes.push_back(w);
  } else {
    static thread_local hipEvent_t ring[32];
    static thread_local int pos = 0, ready = 0;
    if (!ready) {
      for (int i = 0; i < 32; ++i) LMH_CHECK_HIP(hipEventCreateWithFlags(&ring[i], hipEventDisableTiming));
      ready = 1;
    }
    e = ring[pos];
    pos = (pos + 1) & 31;
  }
  LMH_CHECK_HIP(hipEventRecord(e, (hipStream_t)signaler));
  LMH_CHECK_HIP(hipStreamWaitEvent((hipStream_t)waiter, e, 0));
  return LMH_OK;
}

// Record a caller-owned event (lmh_event_create: timing enabled) on `stream`, and make a stream wait for one: the
// recordable forms of hipEventRecord / hipStreamWaitEvent for hand-overs that are not "everything so far" and for the
// timeline marks of bench.py --phases (replayed steps re-record the same event objects).
extern "C" int lmh_event_record(void* event, lmh_stream_t stream) {
  LMH_CHECK_ARG(event != nullptr);
  if (g_rec) {
    PlanNode r{};
    r.kind = NODE_EVENT_RECORD;
    r.st = (hipStream_t)stream;
    r.ev = (hipEvent_t)event;
    g_rec->nodes.push_back(r);
  }
  LMH_CHECK_HIP(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
  return LMH_OK;
}
extern "C" int lmh_stream_wait_event(lmh_stream_t stream, void* event) {
  LMH_CHECK_ARG(event != nullptr);
  if (g_rec) {
    PlanNode w{};
    w.kind = NODE_STREAM_WAIT;
    w.st = (hipStream_t)stream;
    w.ev = (hipEvent_t)event;
    g_rec->nodes.push_back(w);
  }
  LMH_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
  return LMH_OK;
}
extern "C" int lmh_memcpy_d2d(void* dst, const void* src, size_t bytes, lmh_stream_t stream) {
  LMH_CHECK_ARG((dst != nullptr && src != nullptr) || bytes == 0);
  if (bytes == 0) return LMH_OK;
  LMH_CHECK_HIP(lmh_memcpy_d2d_async(dst, src, bytes, (hipStream_t)stream));
  return LMH_OK;
}
extern "C" int lmh_memset(void* ptr, int value, size_t bytes, lmh_stream_t stream) {
  LMH_CHECK_ARG(ptr != nullptr || bytes == 0);
  if (bytes == 0) return LMH_OK;
  LMH_CHECK_HIP(lmh_memset_async(ptr, value, bytes, (hipStream_t)stream));
  return LMH_OK;
}

// ------------------------------------------------------------------------------------------ plan objects ---------
extern "C" int lmh_plan_begin(void) {
  if (g_rec) {
    lmh_set_error("lmh_plan_begin: this thread is already recording a plan");
    return LMH_ERR_INVALID;
  }
  g_rec = new Plan();
  return LMH_OK;
}
// Number of nodes recorded so far by the calling thread (-1 when it is not recording): the host marks the places
// where it has to do something itself between two parts of a plan (gradient all-reduce of a finished bucket).
extern "C" int lmh_plan_position(void) { return g_rec ? (int)g_rec->nodes.size() : -1; }
extern "C" int lmh_plan_recording(void) { return g_rec ? 1 : 0; }

static void plan_finalize(Plan* p) {
  p->arg_ptr.resize(p->arg_off.size() + 1);
  for (size_t i = 0; i < p->arg_off.size(); ++i) p->arg_ptr[i] = p->blob.data() + p->arg_off[i];
  p->finalized = true;
}
extern "C" void* lmh_plan_end(void) {
  Plan* p = g_rec;
  g_rec = nullptr;
  if (!p) {
    lmh_set_error("lmh_plan_end: no recording in progress on this thread");
    return nullptr;
  }
  plan_finalize(p);
  return (void*)p;
}
// Drop the recording in progress (a step raised part-way).
extern "C" void lmh_plan_abort(void) {
  Plan* p = g_rec;
  g_rec = nullptr;
  if (p) {
    for (hipEvent_t e : p->owned) (void)hipEventDestroy(e);
    delete p;
  }
}
extern "C" void lmh_plan_destroy(void* plan) {
  Plan* p = (Plan*)plan;
  if (!p) return;
  for (hipEvent_t e : p->owned) (void)hipEventDestroy(e);
  delete p;
}
extern "C" int lmh_plan_size(void* plan) { return plan ? (int)((Plan*)plan)->nodes.size() : 0; }
// Kernel launches among nodes [first, last) (diagnostics: launches per step).
extern "C" int lmh_plan_kernel_count(void* plan, int first, int last) {
  Plan* p = (Plan*)plan;
  if (!p) return 0;
  if (last < 0 || last > (int)p->nodes.size()) last = (int)p->nodes.size();
  int n = 0;
  for (int i = first < 0 ? 0 : first; i < last; ++i) n += p->nodes[i].kind == NODE_KERNEL;
  return n;
}

// Re-issue nodes [first, last) (last < 0: to the end) on the streams they were recorded on.
extern "C" int lmh_plan_run(void* plan, int first, int last) {
  Plan* p = (Plan*)plan;
  LMH_CHECK_ARG(p != nullptr && p->finalized);
  if (g_rec) {
    lmh_set_error("lmh_plan_run: the calling thread is recording a plan");
    return LMH_ERR_INVALID;
  }
  const int n = (int)p->nodes.size();
  if (last < 0 || last > n) last = n;
  LMH_CHECK_ARG(first >= 0 && first <= last);
  for (int i = first; i < last; ++i) {
    const PlanNode& nd = p->nodes[i];
    switch (nd.kind) {
      case NODE_KERNEL:
        LMH_CHECK_HIP(hipLaunchKernel(nd.fn, nd.grid, nd.block, p->arg_ptr.data() + nd.arg_first, nd.shmem, nd.st));
        break;
      case NODE_MEMSET:
        LMH_CHECK_HIP(hipMemsetAsync(nd.ptr, nd.value, nd.bytes, nd.st));
        break;
      case NODE_MEMCPY:
        LMH_CHECK_HIP(hipMemcpyAsync(nd.ptr, nd.src, nd.bytes, hipMemcpyDeviceToDevice, nd.st));
        break;
      case NODE_EVENT_RECORD:
        LMH_CHECK_HIP(hipEventRecord(nd.ev, nd.st));
        break;
      case NODE_STREAM_WAIT:
        LMH_CHECK_HIP(hipStreamWaitEvent(nd.st, nd.ev, 0));
        break;
    }
  }
  return LMH_OK;
}

// A HIP stream whose kernels may only occupy a subset of the compute units: of every XCD's CUs c = 0..31 those with
// c % period < keep.  Used for the backward-overlap experiment (weight-gradient stream on a fraction of the chip
// beside the data-gradient stream); returns NULL on failure.  Destroy with lmh_stream_destroy.
static lmh_stream_t stream_with_cu_range(int period, int lo, int hi) {
  if (period < 1 || lo < 0 || hi <= lo || hi > period) {
    lmh_set_error("lmh_stream_create_cu_mask: need 0 <= lo < hi <= period");
    return nullptr;
  }
  int dev = 0, ncu = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      ncu <= 0) {
    lmh_set_error("lmh_stream_create_cu_mask: cannot query the device");
    return nullptr;
  }
  const int words = (ncu + 31) / 32;
  std::vector<uint32_t> mask(words, 0u);
  // Measured on MI355X (scripts/probe_cu_mask.py): mask bit i is CU (i / 8) of XCD (i % 8) — a mask that leaves an XCD
  // without any CU is ignored by the runtime (patterns with a power-of-two stride ran at full speed), so the subset is
  // chosen on the per-XCD index and every XCD keeps (hi - lo) / period of its CUs.
  const int xcds = 8;
  for (int i = 0; i < ncu; ++i) {
    const int c = (i / xcds) % period;
    if (c >= lo && c < hi) mask[i >> 5] |= 1u << (i & 31);
  }
  hipStream_t st = nullptr;
  if (hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask.data()) != hipSuccess) {
    lmh_set_error("hipExtStreamCreateWithCUMask failed");
    return nullptr;
  }
  return (lmh_stream_t)st;
}
extern "C" lmh_stream_t lmh_stream_create_cu_mask(int period, int keep) { return stream_with_cu_range(period, 0, keep); }
// ... those with lo <= c % period < hi: two streams with complementary ranges PARTITION the chip (round 5: the proposal /
// RCNN chain on CUs of its own while the convolution streams run on the rest)
extern "C" lmh_stream_t lmh_stream_create_cu_range(int period, int lo, int hi) { return stream_with_cu_range(period, lo, hi); }
extern "C" void lmh_stream_destroy(lmh_stream_t stream) {
  if (stream) (void)hipStreamDestroy((hipStream_t)stream);
}
