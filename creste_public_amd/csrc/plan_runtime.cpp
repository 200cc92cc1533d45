// Python-free deployment entry of libcreste_hip.so: load a launch PLAN exported by creste_public_amd.deploy.export_plan
// and run the RGB-D -> BEV costmap forward from C on caller-owned inputs.
//
// reference: scripts/runtime/compile.py:160-210 traces TraversabilityModel / TerrainNet with torch.jit.trace and saves
// a self-contained TorchScript module that the C++ sister stack loads without Python.  The counterpart here is not a
// traced program but the recorded sequence of C-ABI launches of ONE forward at fixed shapes: a plan file holds
//   * the memory arena (segment sizes) the launches address, every pointer argument as (segment, offset);
//   * the constant blocks (packed BN-folded weights, biases, geometry tables) with their bytes -- no pickle;
//   * the calls: entry-point name + scalar arguments (creste_conv_desc by value with a relocation list);
//   * named inputs (rgbd, p2p) and outputs (the reference's output-dict keys, shape / strides / dtype).
// creste_hip_model_load() allocates the arena, uploads the constants and resolves the pointers; _infer() copies the
// caller's device inputs into the arena and replays the calls on the caller's stream -- optionally through a hipGraph
// captured on the first call (flags & 1).  Results are bit-identical to the Python host path: it IS the same launches.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

namespace creste {
void set_error(const char* fmt, ...);
size_t conv_desc_bytes();            // sizeof(creste_conv_desc) of this build (csrc/conv_igemm.hip)
}
extern "C" int creste_abi_version(void);

namespace {

union PlanArg {
  int i;
  long long l;
  float f;
  double d;
  void* p;
};
struct PlanFn {
  const char* name;
  int nargs;
  const char* sig;                 // one character per argument: i / l / f / d scalars, p pointer, D creste_conv_desc
  int (*call)(const PlanArg*, void*);
};

#include "plan_dispatch.inc"

constexpr uint32_t kNullSeg = 0xffffffffu;
enum ArgKind : uint32_t { K_I32 = 0, K_I64 = 1, K_F32 = 2, K_F64 = 3, K_PTR = 4, K_DESC = 5 };

struct Tensor {
  std::string name;
  uint32_t seg;
  uint64_t off, bytes;
  int dtype, ndim;
  int64_t shape[6], stride[6];
};
struct Call {
  const PlanFn* fn;                // nullptr: a stream-order edge (op = 1 record / 2 wait, args[0].i = event index)
  int op = 0;
  uint32_t stream = 0;             // 0 = the caller's stream, k > 0 = the runtime's own side stream k - 1
  std::vector<PlanArg> args;
};
struct Model {
  std::vector<void*> segs;
  std::vector<uint64_t> seg_bytes;
  std::vector<Tensor> inputs, outputs;
  std::vector<Call> calls;
  std::vector<std::vector<unsigned char>> descs;      // conv descriptors, pointers resolved
  std::string info;
  int flags = 0;
  hipGraphExec_t graph = nullptr;
  hipStream_t graph_stream = nullptr;
  std::vector<hipStream_t> side;   // the streams a pipelined plan's parts run on (the Python path's side streams)
  std::vector<hipEvent_t> events;  // one per recorded fork / join / buffer-ordering edge
  int device = 0;
};

struct Reader {
  FILE* f;
  bool ok = true;
  template <typename T>
  T get() {
    T v{};
    if (ok && fread(&v, sizeof(T), 1, f) != 1) ok = false;
    return v;
  }
  std::string str() {
    const uint32_t n = get<uint32_t>();
    std::string s;
    if (!ok || n > (1u << 20)) { ok = false; return s; }
    s.resize(n);
    if (n && fread(&s[0], 1, n, f) != n) ok = false;
    return s;
  }
  void bytes(void* dst, size_t n) {
    if (ok && n && fread(dst, 1, n, f) != n) ok = false;
  }
};

#define PLAN_FAIL(...)                 \
  do {                                 \
    ::creste::set_error(__VA_ARGS__);  \
    if (f) fclose(f);                  \
    destroy(m);                        \
    return -1;                         \
  } while (0)

void destroy(Model* m) {
  if (!m) return;
  if (m->graph) (void)hipGraphExecDestroy(m->graph);
  if (m->graph_stream) (void)hipStreamDestroy(m->graph_stream);
  for (hipStream_t st : m->side)
    if (st) (void)hipStreamDestroy(st);
  for (hipEvent_t ev : m->events)
    if (ev) (void)hipEventDestroy(ev);
  for (void* p : m->segs)
    if (p) (void)hipFree(p);
  delete m;
}

const PlanFn* find_fn(const std::string& name) {
  for (const PlanFn& fn : kPlanFns)
    if (name == fn.name) return &fn;
  return nullptr;
}

// Replay: every launch on the stream it was recorded on (stream 0 = `s`), the recorded edges as event record / wait pairs.
// A plan traced on one stream has no edges; a pipelined plan (deploy.export_plan of a forward that ran as parts on several
// streams) forks behind the input copies on `s` and joins back onto `s` before its last call returns.
int run_calls(Model* m, hipStream_t s) {
  for (const Call& c : m->calls) {
    hipStream_t st = c.stream == 0 ? s : m->side[c.stream - 1];
    if (c.fn) {
      const int rc = c.fn->call(c.args.data(), (void*)st);
      if (rc != 0) return rc;                         // the entry point has set the error string
      continue;
    }
    hipEvent_t ev = m->events[c.args[0].i];
    const hipError_t e = c.op == 1 ? hipEventRecord(ev, st) : hipStreamWaitEvent(st, ev, 0);
    if (e != hipSuccess) {
      ::creste::set_error("model_infer: stream-order edge failed: %s", hipGetErrorString(e));
      return -2;
    }
  }
  return 0;
}

}  // namespace

extern "C" int creste_hip_model_load(const char* path, int flags, void** handle) {
  FILE* f = nullptr;
  Model* m = nullptr;
  if (!path || !handle) PLAN_FAIL("model_load: null argument");
  f = fopen(path, "rb");
  if (!f) PLAN_FAIL("model_load: cannot open %s", path);
  m = new Model();
  m->flags = flags;
  Reader r{f};
  char magic[12] = {0};
  r.bytes(magic, 12);
  if (!r.ok || memcmp(magic, "CRESTEPLAN\0\0", 12) != 0) PLAN_FAIL("model_load: %s is not a creste plan file", path);
  const uint32_t version = r.get<uint32_t>(), desc_size = r.get<uint32_t>();
  if (version != 3) PLAN_FAIL("model_load: plan format version %u, this library reads 3", version);
  // the recorded arguments are only meaningful to the library generation that recorded them: same C ABI, same
  // descriptor layout (a plan from another ABI would hand creste_conv2d_nhwc a short or mis-laid-out descriptor)
  const uint32_t abi = r.get<uint32_t>();
  if (!r.ok || (int)abi != creste_abi_version())
    PLAN_FAIL("model_load: the plan was exported for C-ABI version %u, this library is version %d: re-export it", abi,
              creste_abi_version());
  if (desc_size != creste::conv_desc_bytes())
    PLAN_FAIL("model_load: the plan's conv descriptor is %u bytes, this library's creste_conv_desc is %zu", desc_size,
              creste::conv_desc_bytes());
  m->info = r.str();
  const uint32_t nseg = r.get<uint32_t>();
  if (!r.ok || nseg > 4096) PLAN_FAIL("model_load: corrupt header");
  if (hipGetDevice(&m->device) != hipSuccess) PLAN_FAIL("model_load: no HIP device");
  m->segs.assign(nseg, nullptr);
  m->seg_bytes.resize(nseg);
  for (uint32_t i = 0; i < nseg; ++i) {
    m->seg_bytes[i] = r.get<uint64_t>();
    if (!r.ok) PLAN_FAIL("model_load: truncated segment table");
    const hipError_t e = hipMalloc(&m->segs[i], m->seg_bytes[i]);
    if (e != hipSuccess) {
      m->segs[i] = nullptr;
      PLAN_FAIL("model_load: hipMalloc(%llu) failed: %s", (unsigned long long)m->seg_bytes[i], hipGetErrorString(e));
    }
  }
  auto resolve = [&](uint32_t seg, uint64_t off, void** out) -> bool {
    if (seg == kNullSeg) { *out = nullptr; return true; }
    if (seg >= nseg || off > m->seg_bytes[seg]) return false;
    *out = (char*)m->segs[seg] + off;
    return true;
  };
  auto read_tensor = [&](Tensor& t) {
    t.name = r.str();
    t.seg = r.get<uint32_t>();
    t.off = r.get<uint64_t>();
    t.bytes = r.get<uint64_t>();
    t.dtype = (int)r.get<uint32_t>();
    t.ndim = (int)r.get<uint32_t>();
    for (int i = 0; i < 6; ++i) t.shape[i] = r.get<int64_t>();
    for (int i = 0; i < 6; ++i) t.stride[i] = r.get<int64_t>();
  };
  const uint32_t nin = r.get<uint32_t>();
  if (!r.ok || nin > 64) PLAN_FAIL("model_load: corrupt input table");
  m->inputs.resize(nin);
  for (Tensor& t : m->inputs) read_tensor(t);
  const uint32_t nout = r.get<uint32_t>();
  if (!r.ok || nout > 256) PLAN_FAIL("model_load: corrupt output table");
  m->outputs.resize(nout);
  for (Tensor& t : m->outputs) read_tensor(t);
  for (const Tensor& t : m->inputs)
    if (t.seg >= nseg || t.off + t.bytes > m->seg_bytes[t.seg]) PLAN_FAIL("model_load: input %s outside the arena", t.name.c_str());
  for (const Tensor& t : m->outputs)
    if (t.seg >= nseg || t.off > m->seg_bytes[t.seg]) PLAN_FAIL("model_load: output %s outside the arena", t.name.c_str());
  // constants: bytes uploaded once
  const uint32_t nconst = r.get<uint32_t>();
  if (!r.ok) PLAN_FAIL("model_load: truncated file");
  std::vector<unsigned char> stage;
  for (uint32_t i = 0; i < nconst; ++i) {
    const uint32_t seg = r.get<uint32_t>();
    const uint64_t off = r.get<uint64_t>(), nb = r.get<uint64_t>();
    if (!r.ok || seg >= nseg || off + nb > m->seg_bytes[seg]) PLAN_FAIL("model_load: constant block %u outside the arena", i);
    stage.resize(nb);
    r.bytes(stage.data(), nb);
    if (!r.ok) PLAN_FAIL("model_load: truncated constant block %u", i);
    if (hipMemcpy((char*)m->segs[seg] + off, stage.data(), nb, hipMemcpyHostToDevice) != hipSuccess)
      PLAN_FAIL("model_load: upload of constant block %u failed", i);
  }
  // calls
  const uint32_t nfn = r.get<uint32_t>();
  if (!r.ok || nfn > 1024) PLAN_FAIL("model_load: corrupt function table");
  std::vector<const PlanFn*> fns(nfn);
  std::vector<int> edge_op(nfn, 0);
  uint32_t nstreams = 1, nevents = 0;
  for (uint32_t i = 0; i < nfn; ++i) {
    const std::string name = r.str();
    fns[i] = find_fn(name);
    if (name == "__record__") edge_op[i] = 1;
    else if (name == "__wait__") edge_op[i] = 2;
    if (!r.ok || (!fns[i] && !edge_op[i])) PLAN_FAIL("model_load: the plan calls %s, which this library does not export", name.c_str());
  }
  const uint32_t ncall = r.get<uint32_t>();
  if (!r.ok || ncall > (1u << 20)) PLAN_FAIL("model_load: corrupt call table");
  m->calls.resize(ncall);
  m->descs.reserve(ncall);
  for (uint32_t c = 0; c < ncall; ++c) {
    const uint32_t fi = r.get<uint32_t>(), sid = r.get<uint32_t>(), na = r.get<uint32_t>();
    if (!r.ok || fi >= nfn || sid >= 16 || (int)na != (fns[fi] ? fns[fi]->nargs : 1))
      PLAN_FAIL("model_load: call %u does not match its entry point", c);
    Call& call = m->calls[c];
    call.fn = fns[fi];
    call.op = edge_op[fi];
    call.stream = sid;
    if (sid + 1 > nstreams) nstreams = sid + 1;
    call.args.resize(na);
    if (!call.fn) {                                   // stream-order edge: one int argument, the event index
      const uint32_t kind = r.get<uint32_t>();
      const int32_t ev = r.get<int32_t>();
      if (!r.ok || kind != K_I32 || ev < 0 || ev >= 4096) PLAN_FAIL("model_load: corrupt stream-order edge in call %u", c);
      call.args[0].l = 0;
      call.args[0].i = ev;
      if ((uint32_t)ev + 1 > nevents) nevents = (uint32_t)ev + 1;
      continue;
    }
    for (uint32_t a = 0; a < na; ++a) {
      const uint32_t kind = r.get<uint32_t>();
      static const char kKindChar[] = {'i', 'l', 'f', 'd', 'p', 'D'};
      if (!r.ok || kind > K_DESC || kKindChar[kind] != call.fn->sig[a])
        PLAN_FAIL("model_load: call %u (%s) argument %u has kind %u, the entry point takes '%c'", c, call.fn->name, a, kind,
                  call.fn->sig[a]);
      PlanArg& v = call.args[a];
      v.l = 0;
      if (kind == K_I32) v.i = r.get<int32_t>();
      else if (kind == K_I64) v.l = r.get<int64_t>();
      else if (kind == K_F32) v.f = r.get<float>();
      else if (kind == K_F64) v.d = r.get<double>();
      else if (kind == K_PTR) {
        const uint32_t seg = r.get<uint32_t>();
        const uint64_t off = r.get<uint64_t>();
        if (!r.ok || !resolve(seg, off, &v.p)) PLAN_FAIL("model_load: call %u argument %u points outside the arena", c, a);
      } else if (kind == K_DESC) {
        m->descs.emplace_back(desc_size);
        std::vector<unsigned char>& d = m->descs.back();
        r.bytes(d.data(), desc_size);
        const uint32_t nrel = r.get<uint32_t>();
        if (!r.ok || nrel > 64) PLAN_FAIL("model_load: corrupt descriptor in call %u", c);
        for (uint32_t k = 0; k < nrel; ++k) {
          const uint32_t foff = r.get<uint32_t>(), seg = r.get<uint32_t>();
          const uint64_t off = r.get<uint64_t>();
          void* p = nullptr;
          if (!r.ok || foff + sizeof(void*) > desc_size || !resolve(seg, off, &p))
            PLAN_FAIL("model_load: bad descriptor relocation in call %u", c);
          memcpy(d.data() + foff, &p, sizeof(void*));
        }
        v.p = d.data();                               // stable: descs was reserved for ncall entries
      } else {
        PLAN_FAIL("model_load: unknown argument kind %u in call %u", kind, c);
      }
    }
  }
  if (!r.ok) PLAN_FAIL("model_load: truncated file");
  m->side.assign(nstreams - 1, nullptr);
  for (hipStream_t& st : m->side)
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { st = nullptr; PLAN_FAIL("model_load: cannot create a side stream"); }
  m->events.assign(nevents, nullptr);
  for (hipEvent_t& ev : m->events)
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { ev = nullptr; PLAN_FAIL("model_load: cannot create an event"); }
  fclose(f);
  *handle = m;
  return 0;
}

extern "C" int creste_hip_model_num_streams(void* handle) { return handle ? 1 + (int)((Model*)handle)->side.size() : -1; }

extern "C" int creste_hip_model_free(void* handle) {
  destroy((Model*)handle);
  return 0;
}

extern "C" const char* creste_hip_model_info(void* handle) { return handle ? ((Model*)handle)->info.c_str() : ""; }

extern "C" int creste_hip_model_num_inputs(void* handle) { return handle ? (int)((Model*)handle)->inputs.size() : -1; }
extern "C" int creste_hip_model_num_outputs(void* handle) { return handle ? (int)((Model*)handle)->outputs.size() : -1; }

static int describe(const Tensor& t, void* base, const char** name, void** ptr, int* dtype, int* ndim, int64_t* shape,
                    int64_t* stride) {
  if (name) *name = t.name.c_str();
  if (ptr) *ptr = (char*)base + t.off;
  if (dtype) *dtype = t.dtype;
  if (ndim) *ndim = t.ndim;
  for (int i = 0; i < 6; ++i) {
    if (shape) shape[i] = t.shape[i];
    if (stride) stride[i] = t.stride[i];
  }
  return 0;
}

extern "C" int creste_hip_model_input(void* handle, int index, const char** name, void** ptr, int* dtype, int* ndim,
                                      int64_t* shape, int64_t* stride) {
  Model* m = (Model*)handle;
  if (!m || index < 0 || index >= (int)m->inputs.size()) { creste::set_error("model_input: bad handle / index"); return -1; }
  const Tensor& t = m->inputs[index];
  return describe(t, m->segs[t.seg], name, ptr, dtype, ndim, shape, stride);
}

extern "C" int creste_hip_model_output(void* handle, int index, const char** name, void** ptr, int* dtype, int* ndim,
                                       int64_t* shape, int64_t* stride) {
  Model* m = (Model*)handle;
  if (!m || index < 0 || index >= (int)m->outputs.size()) { creste::set_error("model_output: bad handle / index"); return -1; }
  const Tensor& t = m->outputs[index];
  return describe(t, m->segs[t.seg], name, ptr, dtype, ndim, shape, stride);
}

// inputs[i]: DEVICE pointer to the i-th input (contiguous, the plan's shape), or NULL when the caller has already
// written it into the arena (creste_hip_model_input's pointer).  Asynchronous on `stream`; outputs are valid once the
// stream has drained and stay valid until the next call.
extern "C" int creste_hip_model_infer(void* handle, const void* const* inputs, int n_inputs, void* stream) {
  Model* m = (Model*)handle;
  if (!m) { creste::set_error("model_infer: null handle"); return -1; }
  if (n_inputs != (int)m->inputs.size()) {
    creste::set_error("model_infer: the plan takes %d inputs, got %d", (int)m->inputs.size(), n_inputs);
    return -1;
  }
  hipStream_t s = (hipStream_t)stream;
  for (int i = 0; i < n_inputs; ++i) {
    if (!inputs || !inputs[i]) continue;
    const Tensor& t = m->inputs[i];
    const hipError_t e = hipMemcpyAsync((char*)m->segs[t.seg] + t.off, inputs[i], t.bytes, hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) { creste::set_error("model_infer: input copy failed: %s", hipGetErrorString(e)); return -2; }
  }
  if (!(m->flags & 1)) return run_calls(m, s);
  if (!m->graph) {                   // capture once, on a stream of our own (the caller's may be the legacy default
    hipGraph_t g = nullptr;            // stream, which cannot capture); the executable graph launches on any stream
    if (!m->graph_stream && hipStreamCreateWithFlags(&m->graph_stream, hipStreamNonBlocking) != hipSuccess) {
      creste::set_error("model_infer: cannot create the capture stream");
      return -2;
    }
    hipError_t e = hipStreamBeginCapture(m->graph_stream, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) { creste::set_error("model_infer: stream capture failed to start: %s", hipGetErrorString(e)); return -2; }
    const int rc = run_calls(m, m->graph_stream);
    e = hipStreamEndCapture(m->graph_stream, &g);
    if (rc != 0 || e != hipSuccess || !g) { if (g) (void)hipGraphDestroy(g); if (rc == 0) creste::set_error("model_infer: graph capture failed"); return rc ? rc : -2; }
    e = hipGraphInstantiate(&m->graph, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) { m->graph = nullptr; creste::set_error("model_infer: hipGraphInstantiate failed: %s", hipGetErrorString(e)); return -2; }
  }
  const hipError_t e = hipGraphLaunch(m->graph, s);
  if (e != hipSuccess) { creste::set_error("model_infer: hipGraphLaunch failed: %s", hipGetErrorString(e)); return -2; }
  return 0;
}

// Synchronous device -> host copy of raw bytes (the plan exporter reads the constant blocks through it; a helper of
// the tooling, not of the data path).
extern "C" int creste_hip_memcpy_d2h(void* dst_host, const void* src_dev, int64_t nbytes) {
  if (!dst_host || !src_dev || nbytes < 0) { creste::set_error("memcpy_d2h: bad args"); return -1; }
  const hipError_t e = hipMemcpy(dst_host, src_dev, (size_t)nbytes, hipMemcpyDeviceToHost);
  if (e != hipSuccess) { creste::set_error("memcpy_d2h: %s", hipGetErrorString(e)); return -2; }
  return 0;
}
