// rabe_amd engine: HIP kernels (gfx950) + the C ABI of include/rabe_hip.h.
//
// Kernel map (one lane = one independent group operation; integer VALU bound, no MFMA):
//   k_fr_op, k_g*_add/neg/mul, k_gt_*            Level E element batches (rabe_bn operator semantics)
//   k_table_build_{g1,g2,gt}[_w16], _g1_wide      window tables of a fixed base (8 / 16 bits; signed up to 27 bits for G1), once per key
//   k_table_mul_{g1,g2}, k_table_pow_gt          fixed-base scalar multiplication / Gt power
//   k_miller                                      one Miller loop per lane (P affine or Jacobian)
//   k_final_exp                                   product of an item's Miller values + ONE final exponentiation (workspace slots)
//   k_ac17_enc_rows / _c0 / _cp                  ac17::cp_encrypt  (src/schemes/ac17/mod.rs:274-376)
//   k_ac17_keygen_rows / _k0                     ac17::cp_keygen   (:191-264)
//   k_ac17_dec_miller (+ k_final_exp)            ac17::cp_decrypt  (:385-430), six independent Miller loops per item
//   k_g2_prepare_lines, k_ac17_dec_miller2       the same with a prepared key: two pairings per lane on one accumulator
//   k_*_c3                                        three cooperating lanes per pairing (small launches)
// There is no CPU fallback anywhere in this file: without a HIP device every entry point fails.
#include <mutex>
#include <set>
#include "engine_internal.h"


void rhip_ktime_begin(rhip_ctx* ctx, const char* name) {
  if (!ctx->timing) return;
  rhip_ctx::Pending p;
  p.name = name;
  if (hipEventCreate(&p.e0) != hipSuccess || hipEventCreate(&p.e1) != hipSuccess) return;
  (void)hipEventRecord(p.e0, ctx->stream);
  ctx->pending.push_back(p);
}
void rhip_ktime_end(rhip_ctx* ctx) {
  if (!ctx->timing || ctx->pending.empty()) return;
  (void)hipEventRecord(ctx->pending.back().e1, ctx->stream);
}
int32_t rhip_fork(rhip_ctx* ctx) {
  if (!ctx->fork[0]) {
    for (int i = 0; i < 2; i++) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->fork[i], hipStreamNonBlocking));
    for (int i = 0; i < 3; i++) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->fork_ev[i], hipEventDisableTiming));
  }
  HIP_TRY(ctx, hipEventRecord(ctx->fork_ev[0], ctx->stream));
  for (int i = 0; i < 2; i++) HIP_TRY(ctx, hipStreamWaitEvent(ctx->fork[i], ctx->fork_ev[0], 0));
  return RHIP_OK;
}
int32_t rhip_join(rhip_ctx* ctx) {
  for (int i = 0; i < 2; i++) {
    HIP_TRY(ctx, hipEventRecord(ctx->fork_ev[1 + i], ctx->fork[i]));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->fork_ev[1 + i], 0));
  }
  return RHIP_OK;
}
int32_t rhip_fail(rhip_ctx* ctx, hipError_t e, const char* what) {
  if (ctx) {
    ctx->err = std::string(what) + ": " + hipGetErrorString(e);
  }
  return RHIP_ERR_HIP;
}
int32_t rhip_ensure_scratch(rhip_ctx* ctx, size_t bytes) {
  if (ctx->scratch_bytes >= bytes) return RHIP_OK;
  if (ctx->scratch) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipFree(ctx->scratch));
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
  }
  HIP_TRY(ctx, hipMalloc(&ctx->scratch, bytes));
  ctx->scratch_bytes = bytes;
  return RHIP_OK;
}

int32_t rhip_ensure_fe_ws(rhip_ctx* ctx, size_t bytes) {
  if (ctx->fe_ws_bytes >= bytes) return RHIP_OK;
  if (ctx->fe_ws) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipFree(ctx->fe_ws));
    ctx->fe_ws = nullptr;
    ctx->fe_ws_bytes = 0;
  }
  HIP_TRY(ctx, hipMalloc(&ctx->fe_ws, bytes));
  ctx->fe_ws_bytes = bytes;
  return RHIP_OK;
}

int32_t rhip_ensure_work(rhip_ctx* ctx, int slot, size_t bytes, void** out) {
  if (slot < 0 || slot >= rhip_ctx::N_WORK) return RHIP_ERR_ARG;
  if (ctx->work_bytes[slot] < bytes) {
    if (ctx->work[slot]) {
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      HIP_TRY(ctx, hipFree(ctx->work[slot]));
      ctx->work[slot] = nullptr;
      ctx->work_bytes[slot] = 0;
    }
    const size_t want = bytes + bytes / 8 + 256;       // a little head-room: batches of slightly different sizes do not re-allocate
    HIP_TRY(ctx, hipMalloc(&ctx->work[slot], want));
    ctx->work_bytes[slot] = want;
  }
  *out = ctx->work[slot];
  return RHIP_OK;
}

static std::string g_create_err;   // diagnostics for a failed rhip_ctx_create (no ctx exists yet)
// live contexts: rhip_ctx_release_before_final_exp stores a pointer to ANOTHER context, which may be destroyed first
static std::mutex g_live_mu;
static std::set<rhip_ctx*> g_live;
static int32_t create_fail(const char* what, hipError_t e) {
  g_create_err = std::string(what) + ": " + hipGetErrorString(e);
  return RHIP_ERR_NO_DEVICE;
}
extern "C" int32_t rhip_ctx_create(int32_t device, rhip_ctx** out) {
  if (!out) return RHIP_ERR_ARG;
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return create_fail("hipGetDeviceCount", e);
  if (n <= 0 || device < 0 || device >= n) {
    g_create_err = "no such HIP device (count=" + std::to_string(n) + ", asked " + std::to_string(device) + ")";
    return RHIP_ERR_NO_DEVICE;
  }
  e = hipSetDevice(device);
  if (e != hipSuccess) return create_fail("hipSetDevice", e);
  rhip_ctx* c = new rhip_ctx();
  c->device = device;
  int cus = 0;
  e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
  if (e != hipSuccess) { delete c; return create_fail("hipDeviceGetAttribute", e); }
  c->n_cu = cus;
  e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
  if (e != hipSuccess) { delete c; return create_fail("hipStreamCreateWithFlags", e); }
  c->stream = c->own_stream;
  if (const char* pm = getenv("RABE_PAIRING_MODE")) {           // A/B runs of whole test suites: 1 / 3 / 6 / 29 as rhip_ctx_set_pairing_mode
    const int m = atoi(pm);
    if (m == 1 || m == 3 || m == 6 || m == 29 || m == 58 || m == 99) c->pairing_mode = m;
  }
  // known answers of the field arithmetic on every SIMD of THIS device before anything is computed with it (bn254/selftest.h)
  if (rhip_device_selftest(c, nullptr, nullptr) != RHIP_OK) {
    g_create_err = c->err;
    (void)hipStreamDestroy(c->own_stream);
    delete c;
    return RHIP_ERR_HIP;
  }
  {
    std::lock_guard<std::mutex> g(g_live_mu);
    g_live.insert(c);
  }
  *out = c;
  return RHIP_OK;
}
extern "C" void rhip_ctx_destroy(rhip_ctx* ctx) {
  if (!ctx) return;
  {
    std::lock_guard<std::mutex> g(g_live_mu);          // nobody keeps waiting for (or releasing) a context that is going away
    g_live.erase(ctx);
    for (rhip_ctx* o : g_live) if (o->fe_waiter == ctx) o->fe_waiter = nullptr;
  }
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  if (ctx->fe_ws) (void)hipFree(ctx->fe_ws);
  if (ctx->fe_started) (void)hipFree(ctx->fe_started);
  for (int i = 0; i < rhip_ctx::N_WORK; i++) if (ctx->work[i]) (void)hipFree(ctx->work[i]);
  for (int i = 0; i < 2; i++) if (ctx->fork[i]) { (void)hipStreamSynchronize(ctx->fork[i]); (void)hipStreamDestroy(ctx->fork[i]); }
  for (int i = 0; i < 3; i++) if (ctx->fork_ev[i]) (void)hipEventDestroy(ctx->fork_ev[i]);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
}
extern "C" int32_t rhip_ctx_set_stream(rhip_ctx* ctx, void* s) {
  if (!ctx) return RHIP_ERR_ARG;
  ctx->stream = s ? (hipStream_t)s : ctx->own_stream;
  return RHIP_OK;
}
extern "C" int32_t rhip_sync(rhip_ctx* ctx) {
  if (!ctx) return RHIP_ERR_ARG;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return RHIP_OK;
}
extern "C" int32_t rhip_ctx_timing(rhip_ctx* ctx, int32_t enable) {
  if (!ctx) return RHIP_ERR_ARG;
  ctx->timing = enable != 0;
  return RHIP_OK;
}
// Drains the recorded launches: writes "kernel_name total_ms launches\n" lines into buf.
extern "C" int32_t rhip_ctx_timing_read(rhip_ctx* ctx, char* buf, size_t len) {
  if (!ctx || !buf || !len) return RHIP_ERR_ARG;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  std::vector<std::string> names;
  std::vector<double> total;
  std::vector<long> count;
  for (auto& p : ctx->pending) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
      size_t k = 0;
      for (; k < names.size(); k++) if (names[k] == p.name) break;
      if (k == names.size()) { names.push_back(p.name); total.push_back(0); count.push_back(0); }
      total[k] += ms;
      count[k] += 1;
    }
    (void)hipEventDestroy(p.e0);
    (void)hipEventDestroy(p.e1);
  }
  ctx->pending.clear();
  std::string out;
  for (size_t k = 0; k < names.size(); k++) {
    char line[256];
    snprintf(line, sizeof line, "%s %.6f %ld\n", names[k].c_str(), total[k], count[k]);
    out += line;
  }
  strncpy(buf, out.c_str(), len - 1);
  buf[len - 1] = 0;
  return RHIP_OK;
}
// the calling thread's current HIP device becomes the context's (HIP's current device is per thread; kernels are launched on the
// context's stream, which belongs to its device): the host layer's worker threads call this once before they work on an engine
extern "C" int32_t rhip_ctx_make_current(rhip_ctx* ctx) {
  NEED(ctx);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  return RHIP_OK;
}
extern "C" const char* rhip_last_error(rhip_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }
extern "C" int32_t rhip_device_info(rhip_ctx* ctx, int32_t* n_cu, char* name, size_t name_len) {
  if (!ctx) return RHIP_ERR_ARG;
  if (n_cu) *n_cu = ctx->n_cu;
  if (name && name_len) {
    hipDeviceProp_t prop;
    HIP_TRY(ctx, hipGetDeviceProperties(&prop, ctx->device));
    strncpy(name, prop.gcnArchName, name_len - 1);
    name[name_len - 1] = 0;
  }
  return RHIP_OK;
}
extern "C" int32_t rhip_malloc(rhip_ctx* ctx, size_t bytes, void** dev) {
  if (!ctx || !dev) return RHIP_ERR_ARG;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipMalloc(dev, bytes ? bytes : 4));
  return RHIP_OK;
}
extern "C" int32_t rhip_free(rhip_ctx* ctx, void* dev) {
  if (!ctx) return RHIP_ERR_ARG;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipFree(dev));
  return RHIP_OK;
}
extern "C" int32_t rhip_upload(rhip_ctx* ctx, void* dev, const void* host, size_t bytes) {
  if (!ctx) return RHIP_ERR_ARG;
  HIP_TRY(ctx, hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return RHIP_OK;
}
extern "C" int32_t rhip_memset_async(rhip_ctx* ctx, void* dev, int32_t byte, size_t bytes) {
  if (!ctx) return RHIP_ERR_ARG;
  if (bytes) HIP_TRY(ctx, hipMemsetAsync(dev, byte, bytes, ctx->stream));
  return RHIP_OK;
}
extern "C" int32_t rhip_download(rhip_ctx* ctx, void* host, const void* dev, size_t bytes) {
  if (!ctx) return RHIP_ERR_ARG;
  HIP_TRY(ctx, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return RHIP_OK;
}

// pinned host memory + copies that are only ordered on the context's stream (rhip_sync waits): lets a caller overlap
// the PCIe traffic of one batch with the kernels of the others
extern "C" int32_t rhip_host_alloc(rhip_ctx* ctx, size_t bytes, void** host) {
  if (!ctx || !host) return RHIP_ERR_ARG;
  HIP_TRY(ctx, hipHostMalloc(host, bytes ? bytes : 4, hipHostMallocDefault));
  return RHIP_OK;
}
extern "C" int32_t rhip_host_free(rhip_ctx* ctx, void* host) {
  if (!ctx) return RHIP_ERR_ARG;
  HIP_TRY(ctx, hipHostFree(host));
  return RHIP_OK;
}
extern "C" int32_t rhip_upload_async(rhip_ctx* ctx, void* dev, const void* host, size_t bytes) {
  if (!ctx) return RHIP_ERR_ARG;
  HIP_TRY(ctx, hipSetDevice(ctx->device));          // callable from a helper thread (the host layer uploads a blob beside its parsing)
  HIP_TRY(ctx, hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, ctx->stream));
  return RHIP_OK;
}
extern "C" int32_t rhip_download_async(rhip_ctx* ctx, void* host, const void* dev, size_t bytes) {
  if (!ctx) return RHIP_ERR_ARG;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
  return RHIP_OK;
}

// work submitted to `ctx` after this call starts only when everything submitted to `other` so far has finished
// Pipelining of two launch sets on two contexts: `waiter`'s stream is held until THIS context's next multi-pairing launch has
// issued its Miller kernel -- from there only the final exponentiation of this context is left, one wave per item, which fills
// the chip only for >= 65 536 items; the waiter's occupancy-flexible kernels (the fixed-base encrypt kernels of the next launch
// set) then run beside it instead of behind it.  One-shot; without a following pairing launch nothing is held.
extern "C" int32_t rhip_ctx_release_before_final_exp(rhip_ctx* ctx, rhip_ctx* waiter) {
  if (!ctx || ctx == waiter) return RHIP_ERR_ARG;
  std::lock_guard<std::mutex> g(g_live_mu);
  if (waiter && !g_live.count(waiter)) return RHIP_ERR_ARG;
  ctx->fe_waiter = waiter;          // NULL withdraws a pending request; rhip_ctx_destroy(waiter) withdraws it too
  ctx->fe_waiter_poll = true;
  ctx->early_release = false;
  return RHIP_OK;
}
// The same hold without the wait for resident final-exponentiation blocks: the waiter goes on as soon as the Miller loops are done.  For a
// waiter whose kernels are small next to the final exponentiation (the host layer's Gt membership checks): the polling kernel of the full
// form occupies the waiter's hardware queue, and a process with more streams than hardware queues may have put ctx's own stream on that
// queue -- measured: the final exponentiation then starts only when the poll gives up (36 ms).
extern "C" int32_t rhip_ctx_release_after_miller(rhip_ctx* ctx, rhip_ctx* waiter) {
  const int32_t rc = rhip_ctx_release_before_final_exp(ctx, waiter);
  if (rc == RHIP_OK) ctx->fe_waiter_poll = false;
  return rc;
}
// The hold for a SMALL launch set on ctx beside the waiter's large one: the waiter's stream goes on as soon as the blocks of ctx's next Miller
// launch are resident (they own their CUs; the waiter's occupancy-flexible kernels take the CUs that are left), instead of when that launch is
// done.  Served by the reduced-radix Miller kernel (k_miller_multi_rr, uniform pair lists); any other launch behaves as under
// rhip_ctx_release_before_final_exp.
extern "C" int32_t rhip_ctx_release_when_miller_resident(rhip_ctx* ctx, rhip_ctx* waiter) {
  const int32_t rc = rhip_ctx_release_before_final_exp(ctx, waiter);
  if (rc == RHIP_OK) ctx->early_release = waiter != nullptr;
  return rc;
}
// a point of a context's stream a HOST thread can wait for (the host layer's helper threads wait for one part's copy, not for the stream)
struct rhip_event { hipEvent_t ev; int device; };
extern "C" int32_t rhip_event_record(rhip_ctx* ctx, rhip_event** out) {
  if (!ctx || !out) return RHIP_ERR_ARG;
  *out = nullptr;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipEvent_t ev;
  HIP_TRY(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  hipError_t e = hipEventRecord(ev, ctx->stream);
  if (e != hipSuccess) { (void)hipEventDestroy(ev); return fail(ctx, e, "hipEventRecord"); }
  *out = new rhip_event{ev, ctx->device};
  return RHIP_OK;
}
extern "C" int32_t rhip_event_wait(rhip_event* ev) {          // blocks the calling thread until the recorded point has been reached; frees the event
  if (!ev) return RHIP_ERR_ARG;
  (void)hipSetDevice(ev->device);
  const hipError_t e = hipEventSynchronize(ev->ev);
  (void)hipEventDestroy(ev->ev);
  delete ev;
  return e == hipSuccess ? RHIP_OK : RHIP_ERR_HIP;
}
extern "C" int32_t rhip_ctx_wait_for(rhip_ctx* ctx, rhip_ctx* other) {
  if (!ctx || !other) return RHIP_ERR_ARG;
  hipEvent_t ev;
  HIP_TRY(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  hipError_t e = hipEventRecord(ev, other->stream);
  if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, ev, 0);
  (void)hipEventDestroy(ev);                 // released by the runtime once the recorded work has completed
  if (e != hipSuccess) return fail(ctx, e, "rhip_ctx_wait_for");
  return RHIP_OK;
}


// ------------------------------------------------------------------------------------------------
// Level E kernels
__global__ void __launch_bounds__(256, RB_MIN_WAVES) k_fr_op(int op, size_t n, const rhip_fr* a, const rhip_fr* b, rhip_fr* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr x = load_fr(a[i].l);
  Fr r;
  if (op == RHIP_FR_NEG) r = neg(x);
  else if (op == RHIP_FR_INV) r = inv(x);
  else if (op == RHIP_FR_POW) {            // `Fr::pow(Fr)` (src/utils/secretsharing/mod.rs:218): x^e, e = b[i] as an integer
    uint32_t e[8];
    ld_scalar(e, b + i);
    r = one<FrParams>();
    for (int w = 7; w >= 0; w--) {
      const uint32_t word = word_sel8(e, w);
      for (int bit = 31; bit >= 0; bit--) {
        r = sqr(r);
        if ((word >> bit) & 1u) r = mul(r, x);
      }
    }
  } else {
    Fr y = load_fr(b[i].l);
    r = (op == RHIP_FR_ADD) ? add(x, y) : (op == RHIP_FR_SUB) ? sub(x, y) : mul(x, y);
  }
  store_fr(out[i].l, r);
}
__global__ void __launch_bounds__(256, RB_MIN_WAVES) k_fr_from_be32(size_t n, const uint8_t* dig, rhip_fr* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t x[8];
  const uint8_t* d = dig + 32 * i;
#pragma unroll
  for (int w = 0; w < 8; w++) {
    const uint8_t* q = d + 28 - 4 * w;   // limb w = bytes [28-4w, 32-4w) big-endian
    x[w] = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | (uint32_t)q[3];
  }
  store_fr(out[i].l, to_mont_reduce256<FrParams>(x));
}
__global__ void __launch_bounds__(256, RB_G1_WAVES) k_g1_add(size_t n, const rhip_g1* a, const rhip_g1* b, rhip_g1* out, int negate_b) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  G1Aff pb = load_g1(b[i].l);
  if (negate_b) pb = aff_neg(pb);
  store_g1(out[i].l, jac_to_aff(jac_add_aff(aff_to_jac(load_g1(a[i].l)), pb)));
}
__global__ void __launch_bounds__(256, RB_MIN_WAVES) k_g1_neg(size_t n, const rhip_g1* a, rhip_g1* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  store_g1(out[i].l, aff_neg(load_g1(a[i].l)));
}
__global__ void __launch_bounds__(256, RB_G1_WAVES) k_g1_mul(size_t n, const rhip_g1* p, const rhip_fr* k, rhip_g1* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t kk[8];
  ld_scalar(kk, k + i);
  store_g1(out[i].l, jac_to_aff(jac_mul_glv_g1(load_g1(p[i].l), kk)));      // any 256-bit k (curve.h)
}
__global__ void __launch_bounds__(256, RB_MIN_WAVES) k_g1_on_curve(size_t n, const rhip_g1* p, uint32_t* ok) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  ok[i] = (wire_words_canonical(p[i].l, 2) && aff_on_curve(load_g1(p[i].l))) ? 1u : 0u;
}
__global__ void __launch_bounds__(128, RB_MIN_WAVES) k_g2_add(size_t n, const rhip_g2* a, const rhip_g2* b, rhip_g2* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  store_g2(out[i].l, jac_to_aff(jac_add_aff(aff_to_jac(load_g2(a[i].l)), load_g2(b[i].l))));
}
__global__ void __launch_bounds__(128, RB_MIN_WAVES) k_g2_neg(size_t n, const rhip_g2* a, rhip_g2* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  store_g2(out[i].l, aff_neg(load_g2(a[i].l)));
}
__global__ void __launch_bounds__(128, RB_MIN_WAVES) k_g2_mul(size_t n, const rhip_g2* p, const rhip_fr* k, rhip_g2* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t kk[8];
  ld_scalar(kk, k + i);
  store_g2(out[i].l, jac_to_aff(jac_mul_binary(load_g2(p[i].l), kk)));
}
__global__ void __launch_bounds__(128, RB_MIN_WAVES) k_g2_on_curve(size_t n, const rhip_g2* p, uint32_t* ok) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  ok[i] = (wire_words_canonical(p[i].l, 4) && aff_on_curve(load_g2(p[i].l))) ? 1u : 0u;
}
__global__ void __launch_bounds__(64, RB_MIN_WAVES) k_gt_mul(size_t n, const rhip_gt* a, const rhip_gt* b, rhip_gt* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  store_gt(out[i].l, fp12_mul(load_gt(a[i].l), load_gt(b[i].l)));
}
// out[i] = prod_{j in [off[i], off[i+1])} a[j]   (1 for an empty segment)
__global__ void __launch_bounds__(64, RB_MIN_WAVES) k_gt_product(size_t n_items, const uint32_t* off, const rhip_gt* a, rhip_gt* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_items) return;
  Fp12 acc = fp12_one();
  for (uint32_t j = off[i]; j < off[i + 1]; j++) {
    Fp12 m = load_gt(a[j].l);
    acc = (j == off[i]) ? m : fp12_mul_fn(acc, m);
  }
  store_gt(out[i].l, acc);
}
__global__ void __launch_bounds__(64, RB_MIN_WAVES) k_gt_inv(size_t n, const rhip_gt* a, rhip_gt* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  store_gt(out[i].l, fp12_inv(load_gt(a[i].l)));
}
__global__ void __launch_bounds__(64, RB_MIN_WAVES) k_gt_pow(size_t n, const rhip_gt* a, const rhip_fr* k, rhip_gt* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t kk[8];
  ld_scalar(kk, k + i);
  store_gt(out[i].l, gt_pow_window(load_gt(a[i].l), kk));
}

// ------------------------------------------------------------------------------------------------
// pairing: Miller loops (one lane per pair) then one final exponentiation per item
// p_kind: 0 = affine canonical rhip_g1 input, 1 = Jacobian Montgomery G1JM input (no inversion was done)
__global__ void __launch_bounds__(64, RB_MIN_WAVES) k_miller(size_t n, const rhip_g1* p, const rhip_g2* q, GtM* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  G1Aff P = load_g1(p[i].l);
  Fp12 f = miller_loop(miller_p_from_aff(P), aff_is_inf(P), load_g2(q[i].l));
  st_gt_m(out + i, f);
}
// out[item] = (mul_in ? mul_in[item] : 1) * FE( prod_{j in [off[item], off[item+1])} mill[j] ), canonical.
// The exponentiation's Fq12 values live in the context's workspace (final_exponentiation_ws, bn254/pairing.h).
// Blocks of four waves: a block owns one CU (see rb_facc_lds4) -- 65 536 items are one block per CU, fewer items leave whole CUs free.
#define RB_FE_BLOCK 256
__global__ void __launch_bounds__(RB_FE_BLOCK, RB_MIN_WAVES) k_final_exp(size_t n_items, const uint32_t* off, uint32_t stride, const GtM* mill,
                                                  const rhip_gt* mul_in, rhip_gt* out, uint32_t* ws_base, size_t ws_stride, uint32_t* started) {
  // rhip_ctx_release_before_final_exp: every block announces that it is resident (the held stream's first kernel waits for the count)
  if (started && threadIdx.x == 0) { atomicAdd(started, 1u); __threadfence(); }
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_items) return;
  const size_t lo = off ? off[i] : i * stride, hi = off ? off[i + 1] : (i + 1) * stride;
  const DevWsT<LdsHomeT<4>> ws{(uint4*)ws_base + i, ws_stride};
  if (lo == hi) ws.st(FE_T0, fp12_one());
  for (size_t j = lo; j < hi; j++) {
    ws.st(j == lo ? FE_T0 : FE_T1, ld_gt_m(mill + j));
    if (j != lo) wsx_mul(ws, FE_T0, FE_T0, false, FE_T1, false);
  }
  final_exponentiation_ws(ws);
  if (mul_in) {
    ws.st(FE_T0, load_gt(mul_in[i].l));
    wsx_mul(ws, FE_T1, FE_T0, false, FE_T1, false);
  }
  store_gt(out[i].l, ws.ld(FE_T1));
}

// ------------------------------------------------------------------------------------------------
// three-lane cooperative pairing kernels (bn254/coop3.h): a wave holds 21 triples (lane 63 idles); the all-gather
// inside a triple is ds_bpermute (__shfl) limb by limb -- structural, so the values never leave registers.
struct DevComm {
  int L;      // role inside the triple
  int base;   // first lane of the triple
  __device__ __forceinline__ Fp g(const Fp& x, int src) const {
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = (uint32_t)__shfl((int)x.v[i], src);
    return r;
  }
  __device__ __forceinline__ Fp2 g(const Fp2& x, int src) const { return Fp2{g(x.c0, src), g(x.c1, src)}; }
  __device__ __forceinline__ Fp6 g(const Fp6& x, int src) const { return Fp6{g(x.a0, src), g(x.a1, src), g(x.a2, src)}; }
  __device__ __forceinline__ Fp4Pair g(const Fp4Pair& x, int src) const { return Fp4Pair{g(x.r0, src), g(x.r1, src)}; }
  __device__ __forceinline__ Fp2Pair g(const Fp2Pair& x, int src) const { return Fp2Pair{g(x.p, src), g(x.q, src)}; }
  template <class T>
  __device__ __forceinline__ void gather(const T& mine, T* out) const {
    out[0] = g(mine, base);
    out[1] = g(mine, base + 1);
    out[2] = g(mine, base + 2);
  }
};
#define C3_TRIPLES_PER_WAVE 21
__device__ __forceinline__ void st_gt_m_third(GtM* p, const Fp12& a, int L) {
  // each lane of the triple stores one third of the (replicated) value: Fq2 coefficients 2L and 2L+1
  const Fp2 x = sel3(L, a.c0.a0, a.c0.a2, a.c1.a1);
  const Fp2 y = sel3(L, a.c0.a1, a.c1.a0, a.c1.a2);
  st_fp2_m(p->l + 32 * L, x);
  st_fp2_m(p->l + 32 * L + 16, y);
}
__global__ void __launch_bounds__(64, RB_MIN_WAVES) k_miller_c3(size_t n, const rhip_g1* p, const rhip_g2* q, GtM* out) {
  const int lane = threadIdx.x;
  const size_t pair = (size_t)blockIdx.x * C3_TRIPLES_PER_WAVE + lane / 3;
  if (lane >= 3 * C3_TRIPLES_PER_WAVE || pair >= n) return;
  DevComm cm{lane % 3, lane - lane % 3};
  G1Aff P = load_g1(p[pair].l);
  Fp12 f = c3_miller_loop(cm, miller_p_from_aff(P), aff_is_inf(P), load_g2(q[pair].l));
  st_gt_m_third(out + pair, f, cm.L);
}
// product of an item's Miller values + cooperative final exponentiation, one triple per item
__global__ void __launch_bounds__(64, RB_MIN_WAVES) k_final_exp_c3(size_t n_items, const uint32_t* off, uint32_t stride, const GtM* mill,
                                                                  const rhip_gt* mul_in, rhip_gt* out) {
  const int lane = threadIdx.x;
  const size_t i = (size_t)blockIdx.x * C3_TRIPLES_PER_WAVE + lane / 3;
  if (lane >= 3 * C3_TRIPLES_PER_WAVE || i >= n_items) return;
  DevComm cm{lane % 3, lane - lane % 3};
  uint32_t lo = off ? off[i] : (uint32_t)(i * stride);
  uint32_t hi = off ? off[i + 1] : (uint32_t)((i + 1) * stride);
  Fp12 f = fp12_one();
  for (uint32_t j = lo; j < hi; j++) {
    Fp12 m = ld_gt_m(mill + j);
    f = (j == lo) ? m : c3_fp12_mul(cm, f, m);
  }
  Fp12 e = c3_final_exponentiation(cm, f);
  if (mul_in) e = c3_fp12_mul(cm, load_gt(mul_in[i].l), e);
  // canonical store, one third per lane
  const Fp2 x = sel3(cm.L, e.c0.a0, e.c0.a2, e.c1.a1);
  const Fp2 y = sel3(cm.L, e.c0.a1, e.c1.a0, e.c1.a2);
  store_fp2(out[i].l + 32 * cm.L, x);
  store_fp2(out[i].l + 32 * cm.L + 16, y);
}

__device__ __forceinline__ void window_scalar(uint32_t k[8], int w, int d) {
#pragma unroll
  for (int i = 0; i < 8; i++) k[i] = 0;
  const uint32_t v = (uint32_t)d << (8 * (w & 3));
  switch (w >> 2) {
    case 0: k[0] = v; break;
    case 1: k[1] = v; break;
    case 2: k[2] = v; break;
    case 3: k[3] = v; break;
    case 4: k[4] = v; break;
    case 5: k[5] = v; break;
    case 6: k[6] = v; break;
    default: k[7] = v; break;
  }
}
__global__ void __launch_bounds__(256, RB_MIN_WAVES) k_table_build_g1(const rhip_g1* base, G1M* tbl) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= TBL_WINDOWS * TBL_DIGITS) return;
  uint32_t k[8];
  window_scalar(k, t / TBL_DIGITS, t % TBL_DIGITS + 1);
  st_g1_m(tbl + t, jac_to_aff(jac_mul_binary(load_g1(base->l), k)));
}
// 16-bit windows for the hot G1 base (AC17's g: 150 fixed-base multiplications per encrypt):
//   T16[w][d-1] = (d * 65536^w) * base = T8[2w][d & 255] + T8[2w+1][d >> 8],  d = 1..65535, w = 0..15   (67 MB)
__global__ void __launch_bounds__(256, RB_G1_WAVES) k_table_build_g1_w16(const G1M* t8, G1M* t16) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)TBL16_WINDOWS * TBL16_DIGITS) return;
  const uint32_t w = (uint32_t)(t / TBL16_DIGITS), d = (uint32_t)(t % TBL16_DIGITS) + 1;
  const uint32_t lo = d & 255u, hi = d >> 8;
  G1Jac acc = jac_inf<Fp>();
  if (lo) acc = jac_add_aff(acc, ld_g1_m(t8 + (2 * w) * TBL_DIGITS + (lo - 1)));
  if (hi) acc = jac_add_aff(acc, ld_g1_m(t8 + (2 * w + 1) * TBL_DIGITS + (hi - 1)));
  st_g1_m(t16 + t, jac_to_aff(acc));
}
// wide table, window i: entry d-1 = (d << (w i)) * base via the 8-bit table, one field inversion per block
__global__ void __launch_bounds__(256, RB_G1_WAVES) k_table_build_g1_wide(const G1M* t8, G1M* out, int w, int i, size_t count) {
  __shared__ uint32_t sh[8][256];
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = t < count;
  if (!active) t = count - 1;
  // m = (t + 1) << (w i), < 2^255
  const uint64_t d = (uint64_t)t + 1;
  const int b = w * i, word = b >> 5, sh_ = b & 31;
  const uint64_t lo = d << sh_;                       // d < 2^27, sh_ < 32: fits 64 bits
  uint32_t m[8];
#pragma unroll
  for (int j = 0; j < 8; j++) m[j] = (j == word) ? (uint32_t)lo : (j == word + 1) ? (uint32_t)(lo >> 32) : 0u;
  const G1Jac r = table_mul_g1(t8, m);
  const Fp zinv = block_batch_inverse_256(sh, r.z);     // never infinity: 0 < m, m is not a multiple of r
  if (active) st_g1_m(out + t, jac_to_aff_with_zinv(r, zinv));
}
// the same for a G2 base (AC17's h_a[j]): T16[w][d-1] = T8[2w][d & 255] + T8[2w+1][d >> 8]   (134 MB)
__global__ void __launch_bounds__(128, RB_MIN_WAVES) k_table_build_g2_w16(const G2M* t8, G2M* t16) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)TBL16_WINDOWS * TBL16_DIGITS) return;
  const uint32_t w = (uint32_t)(t / TBL16_DIGITS), d = (uint32_t)(t % TBL16_DIGITS) + 1;
  const uint32_t lo = d & 255u, hi = d >> 8;
  G2Jac acc = jac_inf<Fp2>();
  if (lo) acc = jac_add_aff(acc, ld_g2_m(t8 + (2 * w) * TBL_DIGITS + (lo - 1)));
  if (hi) acc = jac_add_aff(acc, ld_g2_m(t8 + (2 * w + 1) * TBL_DIGITS + (hi - 1)));
  st_g2_m(t16 + t, jac_to_aff(acc));
}
__global__ void __launch_bounds__(128, RB_MIN_WAVES) k_table_build_g2(const rhip_g2* base, G2M* tbl) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= TBL_WINDOWS * TBL_DIGITS) return;
  uint32_t k[8];
  window_scalar(k, t / TBL_DIGITS, t % TBL_DIGITS + 1);
  st_g2_m(tbl + t, jac_to_aff(jac_mul_binary(load_g2(base->l), k)));
}
__global__ void __launch_bounds__(64, RB_MIN_WAVES) k_table_build_gt(const rhip_gt* base, GtM* tbl) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= TBL_WINDOWS * TBL_DIGITS) return;
  uint32_t k[8];
  window_scalar(k, t / TBL_DIGITS, t % TBL_DIGITS + 1);
  st_gt_m(tbl + t, gt_pow_window(load_gt(base->l), k));
}

// 16-bit windows for Gt powers of public-key constants: T16[w][d-1] = base^(d * 65536^w) = T8[2w][d&255] * T8[2w+1][d>>8]
__global__ void __launch_bounds__(64, RB_MIN_WAVES) k_table_build_gt_w16(const GtM* t8, GtM* t16) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)TBL16_WINDOWS * TBL16_DIGITS) return;
  const uint32_t w = (uint32_t)(t / TBL16_DIGITS), d = (uint32_t)(t % TBL16_DIGITS) + 1;
  const uint32_t lo = d & 255u, hi = d >> 8;
  Fp12 r;
  if (lo && hi) r = fp12_mul(ld_gt_m(t8 + (2 * w) * TBL_DIGITS + (lo - 1)), ld_gt_m(t8 + (2 * w + 1) * TBL_DIGITS + (hi - 1)));
  else if (lo) r = ld_gt_m(t8 + (2 * w) * TBL_DIGITS + (lo - 1));
  else r = ld_gt_m(t8 + (2 * w + 1) * TBL_DIGITS + (hi - 1));
  st_gt_m(t16 + t, r);
}
// out[i] = k[i] * base from the 8-bit (or, when present, 16-bit) window table; one field inversion per block
__global__ void __launch_bounds__(256, RB_G1_WAVES) k_table_mul_g1(const G1M* tbl, size_t n, const rhip_fr* k, rhip_g1* out, int w16) {
  __shared__ uint32_t lds[2 * 8 * 256];
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = i < n;
  if (!active) i = n - 1;
  uint32_t kk[8];
  ld_scalar(kk, k + i);
  const G1Jac r = w16 ? table_mul_g1_w16(tbl, kk) : table_mul_g1(tbl, kk);
  const bool inf = !active || jac_is_inf(r);
  const Fp zinv = block_batch_inverse_n<256>(lds, inf ? one<FpParams>() : r.z);
  if (!active) return;
  store_g1(out[i].l, inf ? aff_inf<Fp>() : jac_to_aff_with_zinv(r, zinv));
}
__global__ void __launch_bounds__(128, RB_G2_WAVES) k_table_mul_g2(const G2M* tbl, size_t n, const rhip_fr* k, rhip_g2* out, int w16) {
  __shared__ uint32_t lds[2 * 8 * 128];
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = i < n;
  if (!active) i = n - 1;
  uint32_t kk[8];
  ld_scalar(kk, k + i);
  store_g2_block128(lds, active, out + i, w16 ? table_mul_g2_w16(tbl, kk) : table_mul_g2(tbl, kk));
}
__global__ void __launch_bounds__(64, RB_MIN_WAVES) k_table_pow_gt(const GtM* tbl, size_t n, const rhip_fr* k, rhip_gt* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t kk[8];
  ld_scalar(kk, k + i);
  bool started = false;
  home_table_pow_gt(started, tbl, kk);
  store_gt(out[i].l, home_result(started));
}

// ------------------------------------------------------------------------------------------------
// AC17
struct rhip_ac17_pk {
  rhip_ctx* ctx;
  rhip_g1_table* g;
  rhip_g2_table* h_a[3];
  rhip_gt_table* e[2];
};


// three Jacobian points -> affine canonical with ONE field inversion per BLOCK (Montgomery's trick twice: over the thread's
// three z's, then over the block's products).  Inactive threads pass active = false.
// The row kernels' form: the first two points are PARKED, as Jacobian Montgomery values, in the row's own output slots (3 x 64 B
// of affine output = exactly 2 x 96 B) by the lane that computed them; the third arrives in registers.  Everything is read into
// registers before the first affine point is stored over the parked data.
__device__ __noinline__ void store3_g1_block_parked(uint32_t* lds, bool active, rhip_g1* out, Fp cx, Fp cy, Fp cz) {
  G1Jac a = jac_inf<Fp>(), b = jac_inf<Fp>();
  const G1Jac c{cx, cy, cz};
  if (active) {
    const G1Jac* park = (const G1Jac*)out;
    a = park[0];
    b = park[1];
  }
  const bool ia = !active || jac_is_inf(a), ib = !active || jac_is_inf(b), ic = !active || jac_is_inf(c);
  Fp za = ia ? one<FpParams>() : a.z, zb = ib ? one<FpParams>() : b.z, zc = ic ? one<FpParams>() : c.z;
  Fp ab = mul(za, zb);
  Fp abc = mul(ab, zc);
  Fp inv_abc = block_batch_inverse_n<RB_ROWS_BLOCK>(lds, abc);
  if (!active) return;
  Fp zc_inv = mul(inv_abc, ab);
  Fp inv_ab = mul(inv_abc, zc);
  Fp zb_inv = mul(inv_ab, za);
  Fp za_inv = mul(inv_ab, zb);
  const G1Aff ra = ia ? aff_inf<Fp>() : jac_to_aff_with_zinv(a, za_inv);
  const G1Aff rb = ib ? aff_inf<Fp>() : jac_to_aff_with_zinv(b, zb_inv);
  const G1Aff rc = ic ? aff_inf<Fp>() : jac_to_aff_with_zinv(c, zc_inv);
  store_g1(out[0].l, ra);
  store_g1(out[1].l, rb);
  store_g1(out[2].l, rc);
}
static_assert(sizeof(G1Jac) == 96 && 2 * sizeof(G1Jac) == 3 * sizeof(rhip_g1), "two parked Jacobian points fill a row's three output records");

// one lane per ciphertext row (items may carry different policies):
//   c[row][l] = g * (s0*A[a][l][0] + s1*A[a][l][1]), l = 0..2, where item = the i with
//   row_off[i] <= row < row_off[i+1] and a = item_A_off[item] + (row - row_off[item]).
__global__ void __launch_bounds__(RB_ROWS_BLOCK, RB_G1_WAVES) k_ac17_enc_rows(const G1M* g_tbl, size_t n_items, size_t total_rows, const rhip_fr* A,
                                                       const uint32_t* item_A_off, const uint32_t* row_off, const rhip_fr* s, rhip_g1* c,
                                                       int w16) {
  __shared__ uint32_t sh[2 * 8 * RB_ROWS_BLOCK];
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = t < total_rows;
  if (!active) t = total_rows - 1;          // inactive lanes shadow the last row (no stores) and still join the block inversion
  // binary search for the item owning row t (row_off is non-decreasing, row_off[n_items] = total_rows)
  size_t lo = 0, hi = n_items;
  while (hi - lo > 1) {
    size_t mid = (lo + hi) >> 1;
    if (row_off[mid] <= t) lo = mid; else hi = mid;
  }
  const size_t item = lo;
  const size_t arow = (size_t)item_A_off[item] + (t - row_off[item]);
  // ONE accumulator is live at a time: the table walk is inlined (the accumulator stays in registers), the finished points of
  // l = 0, 1 are parked in the row's output slots and picked up by the block conversion
  G1Jac* park = (G1Jac*)(c + t * 3);
  G1Jac r = jac_inf<Fp>();
#pragma unroll 1
  for (int l = 0; l < 3; l++) {
    Fr k;
    {
      const Fr s0 = load_fr(s[2 * item].l), s1 = load_fr(s[2 * item + 1].l);
      const Fr a0 = load_fr(A[(arow * 3 + l) * 2].l), a1 = load_fr(A[(arow * 3 + l) * 2 + 1].l);
      Fr t0, t1;
      mul2_inl(t0, t1, s0, a0, s1, a1);          // inlined: no call inside the row loop
      k = add(t0, t1);
    }
    uint32_t kk[8];
    from_mont_inl<FrParams>(kk, k);
    r = (w16 > 16) ? table_mul_g1_wide_inl(g_tbl, kk, w16) : w16 ? table_mul_g1_w16_inl(g_tbl, kk) : table_mul_g1(g_tbl, kk);
    if (l < 2 && active) park[l] = r;
  }
  store3_g1_block_parked(sh, active, c + t * 3, r.x, r.y, r.z);
}
// one lane per (item, j<3): c_0[item][j] = h_a[j] * (s0 | s1 | s0+s1)
__global__ void __launch_bounds__(128, RB_G2_WAVES) k_ac17_enc_c0(const G2M* t0, const G2M* t1, const G2M* t2, size_t n_items,
                                                     const rhip_fr* s, rhip_g2* c0, int w16) {
  __shared__ uint32_t lds[2 * 8 * 128];
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = t < n_items * 3;
  if (!active) t = n_items * 3 - 1;         // shadow lanes join the block inversion, store nothing
  size_t item = t / 3;
  int j = (int)(t % 3);
  uint32_t kk[8];
  if (j < 2) {
    ld_scalar(kk, s + 2 * item + j);
  } else {
    Fr sum = add(load_fr(s[2 * item].l), load_fr(s[2 * item + 1].l));
    from_mont<FrParams>(kk, sum);
  }
  const G2M* tbl = (j == 0) ? t0 : (j == 1) ? t1 : t2;
  store_g2_block128(lds, active, c0 + t, w16 ? table_mul_g2_w16(tbl, kk) : table_mul_g2(tbl, kk));
}
// one lane per item: c_p = e_gh_ka0^s0 * e_gh_ka1^s1 * msg
__global__ void __launch_bounds__(64, RB_MIN_WAVES) k_ac17_enc_cp(const GtM* e0, const GtM* e1, size_t n_items, const rhip_fr* s,
                                                    const rhip_gt* msg, rhip_gt* cp, int w16) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_items) return;
  uint32_t k0[8], k1[8];
  ld_scalar(k0, s + 2 * i);
  ld_scalar(k1, s + 2 * i + 1);
  // msg * e0^s0 * e1^s1 as ONE running product on the lane's home value
  home_put(load_gt(msg[i].l));
  bool started = true;
  if (w16) { home_table_pow_gt_w16(started, e0, k0); home_table_pow_gt_w16(started, e1, k1); }
  else { home_table_pow_gt(started, e0, k0); home_table_pow_gt(started, e1, k1); }
  store_gt(cp[i].l, home_result(true));
}

// keygen: one lane per (item, y <= n_attrs); y == n_attrs is the k_p row.
//   K[y][t]  = g * ((sum_l H[y][l][t]*br_l + sigma_y) * a_t^-1),  K[y][2] = g * (-sigma_y)
//   k_p[t]   = g_k[t] + g * ((sum_l H01[l][t]*br_l + sigma') * a_t^-1), k_p[2] = g_k[2] + g*(-sigma')
__global__ void __launch_bounds__(RB_ROWS_BLOCK, RB_G1_WAVES) k_ac17_keygen_rows(const G1M* g_tbl, const rhip_g1* g_k, const rhip_fr* a_inv, const rhip_fr* b,
                                                          size_t n_items, size_t n_attrs, const rhip_fr* H, const rhip_fr* H01,
                                                          const rhip_fr* r, const rhip_fr* sigma, const rhip_fr* sigma_p,
                                                          rhip_g1* k_out, rhip_g1* kp_out, int w16) {
  __shared__ uint32_t sh[2 * 8 * RB_ROWS_BLOCK];
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t per = n_attrs + 1;
  const bool active = t < n_items * per;
  if (!active) t = n_items * per - 1;
  size_t item = t / per, y = t % per;
  const bool is_kp = (y == n_attrs);
  Fr r0 = load_fr(r[2 * item].l), r1 = load_fr(r[2 * item + 1].l);
  Fr br[3] = {mul(load_fr(b[0].l), r0), mul(load_fr(b[1].l), r1), add(r0, r1)};
  Fr sg = load_fr(is_kp ? sigma_p[item].l : sigma[item * n_attrs + y].l);
  const rhip_fr* Hy = is_kp ? H01 : (H + y * 6);
  rhip_g1* dst = is_kp ? (kp_out + item * 3) : (k_out + (item * n_attrs + y) * 3);
  G1Jac* park = (G1Jac*)dst;
  G1Jac rj = jac_inf<Fp>();
#pragma unroll 1
  for (int tt = 0; tt < 3; tt++) {
    Fr k;
    if (tt < 2) {
      Fr acc = sg;
#pragma unroll 1
      for (int l = 0; l < 3; l++) {
        Fr brl = (l == 0) ? br[0] : (l == 1) ? br[1] : br[2];
        acc = add(acc, mul(load_fr(Hy[l * 2 + tt].l), brl));
      }
      k = mul(acc, load_fr(a_inv[tt].l));
    } else {
      k = neg(sg);
    }
    uint32_t kk[8];
    from_mont<FrParams>(kk, k);
    rj = (w16 > 16) ? table_mul_g1_wide_inl(g_tbl, kk, w16) : w16 ? table_mul_g1_w16_inl(g_tbl, kk) : table_mul_g1(g_tbl, kk);
    if (is_kp) rj = jac_add_aff(rj, load_g1(g_k[tt].l));
    if (tt < 2 && active) park[tt] = rj;
  }
  store3_g1_block_parked(sh, active, dst, rj.x, rj.y, rj.z);
}
// k_0[item][j] = h * (b0 r0 | b1 r1 | r0 + r1)
__global__ void __launch_bounds__(128, RB_G2_WAVES) k_ac17_keygen_k0(const G2M* h_tbl, const rhip_fr* b, size_t n_items, const rhip_fr* r, rhip_g2* k0,
                                                                      int w16) {
  __shared__ uint32_t lds[2 * 8 * 128];
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = t < n_items * 3;
  if (!active) t = n_items * 3 - 1;
  size_t item = t / 3;
  int j = (int)(t % 3);
  Fr r0 = load_fr(r[2 * item].l), r1 = load_fr(r[2 * item + 1].l);
  Fr k = (j == 0) ? mul(load_fr(b[0].l), r0) : (j == 1) ? mul(load_fr(b[1].l), r1) : add(r0, r1);
  uint32_t kk[8];
  from_mont<FrParams>(kk, k);
  store_g2_block128(lds, active, k0 + t, w16 ? table_mul_g2_w16(h_tbl, kk) : table_mul_g2(h_tbl, kk));
}

// decrypt: one lane per (item, i < 6).
//   i < 3 : P = sum_{x in ct_sel} C[x][i],                 Q = k_0[i]      (prod2, :416)
//   i >= 3: P = -(k_p[i-3] + sum_{x in sk_sel} K[x][i-3]), Q = c_0[i-3]    (prod1^-1, :415,418)
// P stays Jacobian (the line is scaled by Z^3 instead of inverting).
__global__ void __launch_bounds__(64, RB_MIN_WAVES) k_ac17_dec_miller(size_t n_items, const rhip_g2* ct_c0, const rhip_g1* ct_c, const uint32_t* ct_row_off,
                                                        const rhip_g2* sk_k0, const rhip_g1* sk_k, const uint32_t* sk_row_off,
                                                        const rhip_g1* sk_kp, const uint32_t* sk_idx, const uint32_t* ct_sel,
                                                        const uint32_t* ct_sel_off, const uint32_t* sk_sel, const uint32_t* sk_sel_off,
                                                        GtM* mill) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_items * 6) return;
  size_t item = t / 6;
  int i = (int)(t % 6);
  const uint32_t sk = sk_idx[item];
  G1Jac acc = jac_inf<Fp>();
  G2Aff Q;
  if (i < 3) {
    const uint32_t base = ct_row_off[item];
    for (uint32_t j = ct_sel_off[item]; j < ct_sel_off[item + 1]; j++)
      acc = jac_add_aff(acc, load_g1(ct_c[(size_t)(base + ct_sel[j]) * 3 + i].l));
    Q = load_g2(sk_k0[(size_t)sk * 3 + i].l);
  } else {
    const int ii = i - 3;
    const uint32_t base = sk_row_off[sk];
    acc = aff_to_jac(load_g1(sk_kp[(size_t)sk * 3 + ii].l));
    for (uint32_t j = sk_sel_off[item]; j < sk_sel_off[item + 1]; j++)
      acc = jac_add_aff(acc, load_g1(sk_k[(size_t)(base + sk_sel[j]) * 3 + ii].l));
    acc = jac_neg(acc);
    Q = load_g2(ct_c0[item * 3 + ii].l);
  }
  Fp12 f = miller_loop(miller_p_from_jac(acc), jac_is_inf(acc), Q);
  st_gt_m(mill + t, f);
}

__global__ void __launch_bounds__(64, RB_MIN_WAVES) k_g2_prepare_lines(size_t n, const rhip_g2* q, LineM* lines, uint8_t* q_inf) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const G2Aff Q = load_g2(q[t].l);
  q_inf[t] = aff_is_inf(Q) ? 1 : 0;
  LineM* out = lines + t * RB_MILLER_LINES;
  if (aff_is_inf(Q)) return;
  // same walk as g2_prepare_lines (bn254/pairing.h), stored as it goes
  G2Hom T{Q.x, Q.y, fp2_one()};
  const G2Aff Qn = aff_neg(Q);
  int n_out = 0;
  auto put = [&](const LineCoeffs& l) {
    uint32_t* p = out[n_out++].l;
    st_fp2_m(p, l.cy); st_fp2_m(p + 16, l.cx); st_fp2_m(p + 32, l.c0);
  };
  for (int i = RB_ATE_NAF_LEN - 2; i >= 0; i--) {
    put(g2hom_double(T));
    const bool pos = (i < 64) && ((RB_ATE_NAF_POS >> i) & 1ull);
    const bool ngt = (i < 64) && ((RB_ATE_NAF_NEG >> i) & 1ull);
    if (pos | ngt) put(g2hom_add(T, pos ? Q : Qn));
  }
  put(g2hom_add(T, g2_frob1(Q)));
  put(g2hom_add(T, aff_neg(g2_frob2(Q))));
}
// LDS parking space of the paired Miller loop (bn254/pairing.h: miller_loop_pair_parked): Fp i of this lane =
// quads 2i, 2i+1 of column `lane`; consecutive lanes are 16 bytes apart, so ds_read_b128 / ds_write_b128 run conflict-free.
struct LdsPark {
  uint4* col;
  __device__ __forceinline__ Fp ld(int i) const {
    const uint4 q0 = col[(2 * i) * 64], q1 = col[(2 * i + 1) * 64];
    Fp r;
    r.v[0] = q0.x; r.v[1] = q0.y; r.v[2] = q0.z; r.v[3] = q0.w;
    r.v[4] = q1.x; r.v[5] = q1.y; r.v[6] = q1.z; r.v[7] = q1.w;
    return r;
  }
  __device__ __forceinline__ void st(int i, const Fp& a) const {
    col[(2 * i) * 64] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
    col[(2 * i + 1) * 64] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
  }
};
// decrypt with a prepared key: one lane per (item, j < 3) runs BOTH pairings of index j on one accumulator
//   A: P = sum_{x in ct_sel} C[x][j],                 Q = k_0[j]   (prepared lines)
//   B: P = -(k_p[j] + sum_{x in sk_sel} K[x][j]),     Q = c_0[j]
// so each doubling step pays one Fq12 squaring instead of two and no G2 arithmetic for A.
__global__ void __launch_bounds__(64, RB_MIN_WAVES) k_ac17_dec_miller2(size_t n_items, const rhip_g2* ct_c0, const rhip_g1* ct_c,
                                                                      const uint32_t* ct_row_off, const LineM* sk_lines, const uint8_t* sk_qinf,
                                                                      const rhip_g1* sk_k, const uint32_t* sk_row_off, const rhip_g1* sk_kp,
                                                                      const uint32_t* sk_idx, const uint32_t* ct_sel, const uint32_t* ct_sel_off,
                                                                      const uint32_t* sk_sel, const uint32_t* sk_sel_off, GtM* mill) {
  __shared__ uint4 park[2 * PK_FPS][64];   // 32 KB: the block is one wave and owns a quarter of the CU's LDS
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_items * 3) return;
  const size_t item = t / 3;
  const int j = (int)(t % 3);
  const uint32_t sk = sk_idx[item];
  G1Jac pa = jac_inf<Fp>();
  {
    const uint32_t base = ct_row_off[item];
    for (uint32_t x = ct_sel_off[item]; x < ct_sel_off[item + 1]; x++)
      pa = jac_add_aff(pa, load_g1(ct_c[(size_t)(base + ct_sel[x]) * 3 + j].l));
  }
  G1Jac pb = aff_to_jac(load_g1(sk_kp[(size_t)sk * 3 + j].l));
  {
    const uint32_t base = sk_row_off[sk];
    for (uint32_t x = sk_sel_off[item]; x < sk_sel_off[item + 1]; x++)
      pb = jac_add_aff(pb, load_g1(sk_k[(size_t)(base + sk_sel[x]) * 3 + j].l));
    pb = jac_neg(pb);
  }
  const size_t lj = (size_t)sk * 3 + j;
  const bool skip_a = jac_is_inf(pa) || sk_qinf[lj];
  const G2Aff QB = load_g2(ct_c0[item * 3 + j].l);
  const bool skip_b = jac_is_inf(pb) || aff_is_inf(QB);
  const LdsPark pk{&park[0][threadIdx.x]};
  pk_st_p(pk, PK_PA, miller_p_from_jac(pa));
  pk_st_p(pk, PK_PB, miller_p_from_jac(pb));
  pk_st2(pk, PK_QB, QB.x);
  pk_st2(pk, PK_QB + 2, QB.y);
  Fp12 f = miller_loop_pair_parked(pk, skip_a, DevLineLoad{sk_lines + lj * RB_MILLER_LINES}, skip_b);
  st_gt_m(mill + t, f);
}

// three-lane variant of k_ac17_dec_miller: triple = (item, i < 6); the (short) G1 row sums are replicated in the
// three lanes, the Miller loop is cooperative.
__global__ void __launch_bounds__(64, RB_MIN_WAVES) k_ac17_dec_miller_c3(size_t n_items, const rhip_g2* ct_c0, const rhip_g1* ct_c,
                                                                        const uint32_t* ct_row_off, const rhip_g2* sk_k0, const rhip_g1* sk_k,
                                                                        const uint32_t* sk_row_off, const rhip_g1* sk_kp, const uint32_t* sk_idx,
                                                                        const uint32_t* ct_sel, const uint32_t* ct_sel_off, const uint32_t* sk_sel,
                                                                        const uint32_t* sk_sel_off, GtM* mill) {
  const int lane = threadIdx.x;
  const size_t t = (size_t)blockIdx.x * C3_TRIPLES_PER_WAVE + lane / 3;
  if (lane >= 3 * C3_TRIPLES_PER_WAVE || t >= n_items * 6) return;
  DevComm cm{lane % 3, lane - lane % 3};
  size_t item = t / 6;
  int i = (int)(t % 6);
  const uint32_t sk = sk_idx[item];
  G1Jac acc = jac_inf<Fp>();
  G2Aff Q;
  if (i < 3) {
    const uint32_t base = ct_row_off[item];
    for (uint32_t j = ct_sel_off[item]; j < ct_sel_off[item + 1]; j++)
      acc = jac_add_aff(acc, load_g1(ct_c[(size_t)(base + ct_sel[j]) * 3 + i].l));
    Q = load_g2(sk_k0[(size_t)sk * 3 + i].l);
  } else {
    const int ii = i - 3;
    const uint32_t base = sk_row_off[sk];
    acc = aff_to_jac(load_g1(sk_kp[(size_t)sk * 3 + ii].l));
    for (uint32_t j = sk_sel_off[item]; j < sk_sel_off[item + 1]; j++)
      acc = jac_add_aff(acc, load_g1(sk_k[(size_t)(base + sk_sel[j]) * 3 + ii].l));
    acc = jac_neg(acc);
    Q = load_g2(ct_c0[item * 3 + ii].l);
  }
  Fp12 f = c3_miller_loop(cm, miller_p_from_jac(acc), jac_is_inf(acc), Q);
  st_gt_m_third(mill + t, f, cm.L);
}

// ------------------------------------------------------------------------------------------------
// integer-multiply issue-rate calibration (roofline denominator)
template <int VARIANT>
__global__ void __launch_bounds__(256, RB_MIN_WAVES) k_calibrate(uint32_t iters, uint32_t seed, uint32_t* sink) {
  uint32_t a = seed + threadIdx.x * 2654435761u, b = seed ^ (blockIdx.x * 40503u + 77u);
  uint64_t x0 = a, x1 = b, x2 = a ^ b, x3 = a + b, x4 = a * 3u, x5 = b * 5u, x6 = a * 7u, x7 = b * 9u;
  if (VARIANT == 5) {
    // Montgomery multiplication throughput (the unit the kernels are really made of)
    Fp u, v;
#pragma unroll
    for (int i = 0; i < 8; i++) { u.v[i] = a + i; v.v[i] = b + 3 * i; }
    u.v[7] &= 0x0fffffffu; v.v[7] &= 0x0fffffffu;
    for (uint32_t it = 0; it < iters; it++) {
      u = mul_inl(u, v);
      v = mul_inl(v, u);
    }
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o ^= u.v[i] ^ v.v[i];
    if (o == 0x12345678u) sink[0] = o;
    return;
  }
  double d0 = (double)a, d1 = (double)b, d2 = 1.5, d3 = 2.5, d4 = 3.5, d5 = 4.5, d6 = 5.5, d7 = 6.5;
  const double dm = 1.0000001, da = 0.5;
  for (uint32_t it = 0; it < iters; it++) {
#define RB8(stmt_) stmt_(x0) stmt_(x1) stmt_(x2) stmt_(x3) stmt_(x4) stmt_(x5) stmt_(x6) stmt_(x7)
#define RB_MAD(x) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b) : "vcc");
#define RB_MULLO(x) { uint32_t lo_ = (uint32_t)x; asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo_) : "v"(a)); x = lo_; }
#define RB_MULHI(x) { uint32_t lo_ = (uint32_t)x; asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(lo_) : "v"(a)); x = lo_; }
#define RB_ADD(x) { uint32_t lo_ = (uint32_t)x; asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo_) : "v"(a)); x = lo_; }
#define RB_ADD64(x) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(x) : "v"(x1));
    if (VARIANT == 0) {
      // ONE asm statement per eight mads: hipcc pads every statement boundary with a wait state (tools/ubench_mac.hip), which a
      // statement per instruction turns into a 10 % lower "peak" (30-32 instead of 35 TMAC32/s)
#define RB_MAD8 asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_mad_u64_u32 %2, vcc, %8, %9, %2\n\t" \
                             "v_mad_u64_u32 %3, vcc, %8, %9, %3\n\tv_mad_u64_u32 %4, vcc, %8, %9, %4\n\tv_mad_u64_u32 %5, vcc, %8, %9, %5\n\t" \
                             "v_mad_u64_u32 %6, vcc, %8, %9, %6\n\tv_mad_u64_u32 %7, vcc, %8, %9, %7"                                   \
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
      RB_MAD8 RB_MAD8 RB_MAD8 RB_MAD8
    }
    if (VARIANT == 1) { RB8(RB_MULLO) RB8(RB_MULLO) RB8(RB_MULLO) RB8(RB_MULLO) }
    if (VARIANT == 2) { RB8(RB_ADD) RB8(RB_ADD) RB8(RB_ADD) RB8(RB_ADD) }
    if (VARIANT == 3) {
#define RB_FMA(d) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d) : "v"(dm), "v"(da));
      RB_FMA(d0) RB_FMA(d1) RB_FMA(d2) RB_FMA(d3) RB_FMA(d4) RB_FMA(d5) RB_FMA(d6) RB_FMA(d7)
      RB_FMA(d0) RB_FMA(d1) RB_FMA(d2) RB_FMA(d3) RB_FMA(d4) RB_FMA(d5) RB_FMA(d6) RB_FMA(d7)
      RB_FMA(d0) RB_FMA(d1) RB_FMA(d2) RB_FMA(d3) RB_FMA(d4) RB_FMA(d5) RB_FMA(d6) RB_FMA(d7)
      RB_FMA(d0) RB_FMA(d1) RB_FMA(d2) RB_FMA(d3) RB_FMA(d4) RB_FMA(d5) RB_FMA(d6) RB_FMA(d7)
    }
    if (VARIANT == 4) { RB8(RB_ADD64) RB8(RB_ADD64) RB8(RB_ADD64) RB8(RB_ADD64) }
    if (VARIANT == 6) { RB8(RB_MULHI) RB8(RB_MULHI) RB8(RB_MULHI) RB8(RB_MULHI) }
  }
  uint64_t o = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
  double dd = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7;
  if (o == 0x123456789abcull || dd == 1.2345) sink[0] = (uint32_t)o;
}

extern "C" int32_t rhip_calibrate_mad(rhip_ctx* ctx, int32_t variant, uint32_t iters, double* ms, double* n_ops) {
  if (!ctx || !ms || !n_ops) return RHIP_ERR_ARG;
  uint32_t* sink = nullptr;
  HIP_TRY(ctx, hipMalloc((void**)&sink, 64));
  const unsigned blocks = (unsigned)ctx->n_cu * 8, bs = 256;
  hipEvent_t e0, e1;
  HIP_TRY(ctx, hipEventCreate(&e0));
  HIP_TRY(ctx, hipEventCreate(&e1));
  for (int rep = 0; rep < 2; rep++) {   // first pass warms up
    HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
    switch (variant) {
      case 0: hipLaunchKernelGGL(k_calibrate<0>, dim3(blocks), dim3(bs), 0, ctx->stream, iters, 12345u, sink); break;
      case 1: hipLaunchKernelGGL(k_calibrate<1>, dim3(blocks), dim3(bs), 0, ctx->stream, iters, 12345u, sink); break;
      case 2: hipLaunchKernelGGL(k_calibrate<2>, dim3(blocks), dim3(bs), 0, ctx->stream, iters, 12345u, sink); break;
      case 3: hipLaunchKernelGGL(k_calibrate<3>, dim3(blocks), dim3(bs), 0, ctx->stream, iters, 12345u, sink); break;
      case 4: hipLaunchKernelGGL(k_calibrate<4>, dim3(blocks), dim3(bs), 0, ctx->stream, iters, 12345u, sink); break;
      case 5: hipLaunchKernelGGL(k_calibrate<5>, dim3(blocks), dim3(bs), 0, ctx->stream, iters, 12345u, sink); break;
      case 6: hipLaunchKernelGGL(k_calibrate<6>, dim3(blocks), dim3(bs), 0, ctx->stream, iters, 12345u, sink); break;
      default: (void)hipFree(sink); return RHIP_ERR_ARG;
    }
    LAUNCH_CHECK(ctx, "k_calibrate");
    HIP_TRY(ctx, hipEventRecord(e1, ctx->stream));
    HIP_TRY(ctx, hipEventSynchronize(e1));
  }
  float t = 0;
  HIP_TRY(ctx, hipEventElapsedTime(&t, e0, e1));
  *ms = t;
  const double per_iter = (variant == 5) ? 2.0 : 32.0;   // variant 5 counts Fp multiplications
  *n_ops = (double)blocks * bs * (double)iters * per_iter;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(sink);
  return RHIP_OK;
}

// ------------------------------------------------------------------------------------------------
// C ABI: Level E

extern "C" int32_t rhip_fr_op(rhip_ctx* ctx, int32_t op, size_t n, const rhip_fr* a, const rhip_fr* b, rhip_fr* out) {
  NEED(ctx);
  if (op < 0 || op > RHIP_FR_POW) return RHIP_ERR_ARG;
  if (!n) return RHIP_OK;
  KLAUNCH(ctx, "k_fr_op", k_fr_op, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream, op, n, a, b, out);
  return RHIP_OK;
}
extern "C" int32_t rhip_fr_from_be32_reduce(rhip_ctx* ctx, size_t n, const uint8_t* dig, rhip_fr* out) {
  NEED(ctx);
  if (!n) return RHIP_OK;
  KLAUNCH(ctx, "k_fr_from_be32", k_fr_from_be32, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream, n, dig, out);
  return RHIP_OK;
}
extern "C" int32_t rhip_g1_add(rhip_ctx* ctx, size_t n, const rhip_g1* a, const rhip_g1* b, rhip_g1* out) {
  NEED(ctx);
  if (!n) return RHIP_OK;
  KLAUNCH(ctx, "k_g1_add", k_g1_add, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream, n, a, b, out, 0);
  return RHIP_OK;
}
extern "C" int32_t rhip_g1_neg(rhip_ctx* ctx, size_t n, const rhip_g1* a, rhip_g1* out) {
  NEED(ctx);
  if (!n) return RHIP_OK;
  KLAUNCH(ctx, "k_g1_neg", k_g1_neg, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream, n, a, out);
  return RHIP_OK;
}
extern "C" int32_t rhip_g1_mul(rhip_ctx* ctx, size_t n, const rhip_g1* p, const rhip_fr* k, rhip_g1* out) {
  NEED(ctx);
  if (!n) return RHIP_OK;
  KLAUNCH(ctx, "k_g1_mul", k_g1_mul, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream, n, p, k, out);
  return RHIP_OK;
}
extern "C" int32_t rhip_g1_on_curve(rhip_ctx* ctx, size_t n, const rhip_g1* p, uint32_t* ok) {
  NEED(ctx);
  if (!n) return RHIP_OK;
  KLAUNCH(ctx, "k_g1_on_curve", k_g1_on_curve, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream, n, p, ok);
  return RHIP_OK;
}
extern "C" int32_t rhip_g2_add(rhip_ctx* ctx, size_t n, const rhip_g2* a, const rhip_g2* b, rhip_g2* out) {
  NEED(ctx);
  if (!n) return RHIP_OK;
  KLAUNCH(ctx, "k_g2_add", k_g2_add, dim3(blocks_for(n, 128)), dim3(128), 0, ctx->stream, n, a, b, out);
  return RHIP_OK;
}
extern "C" int32_t rhip_g2_neg(rhip_ctx* ctx, size_t n, const rhip_g2* a, rhip_g2* out) {
  NEED(ctx);
  if (!n) return RHIP_OK;
  KLAUNCH(ctx, "k_g2_neg", k_g2_neg, dim3(blocks_for(n, 128)), dim3(128), 0, ctx->stream, n, a, out);
  return RHIP_OK;
}
extern "C" int32_t rhip_g2_mul(rhip_ctx* ctx, size_t n, const rhip_g2* p, const rhip_fr* k, rhip_g2* out) {
  NEED(ctx);
  if (!n) return RHIP_OK;
  KLAUNCH(ctx, "k_g2_mul", k_g2_mul, dim3(blocks_for(n, 128)), dim3(128), 0, ctx->stream, n, p, k, out);
  return RHIP_OK;
}
extern "C" int32_t rhip_g2_on_curve(rhip_ctx* ctx, size_t n, const rhip_g2* p, uint32_t* ok) {
  NEED(ctx);
  if (!n) return RHIP_OK;
  KLAUNCH(ctx, "k_g2_on_curve", k_g2_on_curve, dim3(blocks_for(n, 128)), dim3(128), 0, ctx->stream, n, p, ok);
  return RHIP_OK;
}
extern "C" int32_t rhip_gt_mul(rhip_ctx* ctx, size_t n, const rhip_gt* a, const rhip_gt* b, rhip_gt* out) {
  NEED(ctx);
  if (!n) return RHIP_OK;
  KLAUNCH(ctx, "k_gt_mul", k_gt_mul, dim3(blocks_for(n, 64)), dim3(64), 0, ctx->stream, n, a, b, out);
  return RHIP_OK;
}
extern "C" int32_t rhip_gt_product(rhip_ctx* ctx, size_t n_items, const uint32_t* off, const rhip_gt* a, rhip_gt* out) {
  NEED(ctx);
  if (!n_items) return RHIP_OK;
  KLAUNCH(ctx, "k_gt_product", k_gt_product, dim3(blocks_for(n_items, 64)), dim3(64), 0, ctx->stream, n_items, off, a, out);
  return RHIP_OK;
}
extern "C" int32_t rhip_gt_inv(rhip_ctx* ctx, size_t n, const rhip_gt* a, rhip_gt* out) {
  NEED(ctx);
  if (!n) return RHIP_OK;
  KLAUNCH(ctx, "k_gt_inv", k_gt_inv, dim3(blocks_for(n, 64)), dim3(64), 0, ctx->stream, n, a, out);
  return RHIP_OK;
}
extern "C" int32_t rhip_gt_pow(rhip_ctx* ctx, size_t n, const rhip_gt* a, const rhip_fr* k, rhip_gt* out) {
  NEED(ctx);
  if (!n) return RHIP_OK;
  KLAUNCH(ctx, "k_gt_pow", k_gt_pow, dim3(blocks_for(n, 64)), dim3(64), 0, ctx->stream, n, a, k, out);
  return RHIP_OK;
}
// Which pairing kernels a launch uses.  Measured on MI355X (profiles/README.md): per wave the three-lane kernels are
// 1.3x faster (Miller 8.4 ms vs 11.1 ms, final exponentiation 8.5 vs 11.6) but cost 2.3x the SIMD time, and at one
// wave per SIMD (512 registers of replicated state) more than 21 504 pairs need a second round.  So "auto" uses them
// only for small launches, where latency is all that matters; throughput-sized batches keep one lane per pairing.
// first kernel on a stream released by rhip_ctx_release_before_final_exp: one wave that waits -- for a bounded number of polls, so
// that nothing can hang on it -- until `target` final-exponentiation waves of the other context are resident.  Those waves need a
// whole SIMD each; released without this, the held stream's many small blocks fill every SIMD's register file first and the final
// exponentiation only starts when they are done (measured: no overlap at all).
__global__ void __launch_bounds__(64) k_wait_resident(const uint32_t* started, uint32_t target, uint32_t max_polls) {
  if (threadIdx.x != 0) return;
  for (uint32_t p = 0; p < max_polls; p++) {
    if (__atomic_load_n(started, __ATOMIC_RELAXED) >= target) break;
    __builtin_amdgcn_s_sleep(64);
  }
}
// A pending rhip_ctx_release_before_final_exp request is consumed here: the waiter's stream continues behind everything ctx's stream holds so far
// and -- in the polling form -- once `blocks` blocks (at most one per CU) of the kernel launched next have announced themselves through *started.
int32_t rhip_take_waiter(rhip_ctx* ctx, size_t blocks, uint32_t** started_out) {
  *started_out = nullptr;
  ctx->early_release = false;
  std::unique_lock<std::mutex> live(g_live_mu);          // the waiter cannot be destroyed while its stream is being touched
  rhip_ctx* w = ctx->fe_waiter;
  ctx->fe_waiter = nullptr;
  if (w && !g_live.count(w)) w = nullptr;
  if (!w) return RHIP_OK;
  if (!ctx->fe_started) HIP_TRY(ctx, hipMalloc((void**)&ctx->fe_started, 256));
  uint32_t* started = (uint32_t*)ctx->fe_started;
  HIP_TRY(ctx, hipMemsetAsync(started, 0, 4, ctx->stream));
  hipEvent_t ev;
  HIP_TRY(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  hipError_t e = hipEventRecord(ev, ctx->stream);                  // everything before the launch is done
  if (e == hipSuccess) e = hipStreamWaitEvent(w->stream, ev, 0);
  (void)hipEventDestroy(ev);
  if (e != hipSuccess) return fail(ctx, e, "rhip_ctx_release_before_final_exp");
  // ... and the launch's waves are resident (at most as many as there are SIMDs); ~2 ms of polling at most
  const uint32_t target = (uint32_t)(blocks < (size_t)ctx->n_cu ? blocks : (size_t)ctx->n_cu);
  if (ctx->fe_waiter_poll) {
    hipLaunchKernelGGL(k_wait_resident, dim3(1), dim3(64), 0, w->stream, (const uint32_t*)started, target, 20000u);
    *started_out = started;
  }
  return RHIP_OK;
}
int32_t rhip_launch_final_exp(rhip_ctx* ctx, size_t n_items, const uint32_t* off, uint32_t stride, const GtM* mill, const rhip_gt* mul_in,
                                rhip_gt* out) {
  const size_t lanes = (n_items + 63) / 64 * 64;
  int32_t rc = ensure_fe_ws(ctx, lanes * FE_SLOTS * 96 * sizeof(uint32_t));
  if (rc) return rc;
  uint32_t* started = nullptr;
  const size_t blocks = blocks_for(n_items, RB_FE_BLOCK);
  rc = rhip_take_waiter(ctx, blocks, &started);
  if (rc) return rc;
  // the six-lane kernel (engine_coop.hip) for launches that leave most of the chip idle: the same values, a chain six times shorter
  if (rhip_use_c6(ctx, n_items, 0)) return rhip_launch_final_exp_c6(ctx, n_items, off, stride, mill, mul_in, out, started);
  // the reduced-radix kernel (engine_rr.hip) for the launches that fill the chip: the same chain on 9 x 29-bit limbs
  static const int rr_fe = getenv("RABE_RR_FE") ? atoi(getenv("RABE_RR_FE")) : 1;
  if (rr_fe && rhip_use_rr(ctx)) return rhip_launch_final_exp_rr(ctx, n_items, off, stride, mill, mul_in, out, started);
  KLAUNCH(ctx, "k_final_exp", k_final_exp, dim3(blocks), dim3(RB_FE_BLOCK), 0, ctx->stream, n_items, off, stride, mill, mul_in, out,
          (uint32_t*)ctx->fe_ws, lanes, started);
  return RHIP_OK;
}
bool rhip_use_c3(const rhip_ctx* ctx, size_t n_pairs) {
  const int mode = rhip_mode(ctx);
  if (mode == 1 || mode == 6 || mode == 29 || mode == 58) return false;
  if (mode == 3) return true;
  return n_pairs * 3 <= (size_t)ctx->n_cu * 4 * 63 / 4;      // at most a quarter of the SIMDs busy with one lane each
}
extern "C" int32_t rhip_ctx_set_pairing_mode(rhip_ctx* ctx, int32_t mode) {
  if (!ctx || (mode != 0 && mode != 1 && mode != 3 && mode != 6 && mode != 29 && mode != 58 && mode != 99)) return RHIP_ERR_ARG;
  ctx->pairing_mode = mode;
  return RHIP_OK;
}
extern "C" int32_t rhip_pairing_product(rhip_ctx* ctx, size_t n_items, const uint32_t* off, size_t n_pairs, const rhip_g1* p,
                                        const rhip_g2* q, rhip_gt* out) {
  NEED(ctx);
  if (!n_items) return RHIP_OK;
  int32_t rc = ensure_scratch(ctx, (n_pairs ? n_pairs : 1) * sizeof(GtM));
  if (rc) return rc;
  GtM* mill = (GtM*)ctx->scratch;
  if (rhip_use_c3(ctx, n_pairs)) {
    if (n_pairs)
      KLAUNCH(ctx, "k_miller_c3", k_miller_c3, dim3(blocks_for(n_pairs, C3_TRIPLES_PER_WAVE)), dim3(64), 0, ctx->stream, n_pairs, p, q, mill);
    KLAUNCH(ctx, "k_final_exp_c3", k_final_exp_c3, dim3(blocks_for(n_items, C3_TRIPLES_PER_WAVE)), dim3(64), 0, ctx->stream, n_items, off, 1u,
            (const GtM*)mill, (const rhip_gt*)nullptr, out);
    return RHIP_OK;
  }
  if (n_pairs) {
    KLAUNCH(ctx, "k_miller", k_miller, dim3(blocks_for(n_pairs, 64)), dim3(64), 0, ctx->stream, n_pairs, p, q, mill);
  }
  return launch_final_exp(ctx, n_items, off, 1u, (const GtM*)mill,
                     (const rhip_gt*)nullptr, out);
}
extern "C" int32_t rhip_pairing(rhip_ctx* ctx, size_t n, const rhip_g1* p, const rhip_g2* q, rhip_gt* out) {
  return rhip_pairing_product(ctx, n, nullptr, n, p, q, out);
}

// ------------------------------------------------------------------------------------------------
// Host-value forms of the Level E operators (one element, host pointers): what an operator-overloading `rabe_bn`
// replacement binds.  Upload, one launch, download -- for parity runs, not throughput.
struct HostOpBuf {
  rhip_ctx* ctx;
  uint8_t* d = nullptr;
  explicit HostOpBuf(rhip_ctx* c) : ctx(c) {}
  ~HostOpBuf() { if (d) (void)hipFree(d); }
  int32_t init(size_t bytes) { HIP_TRY(ctx, hipMalloc((void**)&d, bytes)); return RHIP_OK; }
  int32_t put(size_t off, const void* src, size_t n) { HIP_TRY(ctx, hipMemcpyAsync(d + off, src, n, hipMemcpyHostToDevice, ctx->stream)); return RHIP_OK; }
  int32_t get(void* dst, size_t off, size_t n) {
    HIP_TRY(ctx, hipMemcpyAsync(dst, d + off, n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RHIP_OK;
  }
};
#define HOSTOP_BEGIN(total)                 \
  NEED(ctx);                                \
  HostOpBuf hb(ctx);                        \
  int32_t rc = hb.init(total);              \
  if (rc) return rc;
#define HOSTOP_TRY(x) do { rc = (x); if (rc) return rc; } while (0)
extern "C" int32_t rhip_host_fr_op(rhip_ctx* ctx, int32_t op, const rhip_fr* a, const rhip_fr* b, rhip_fr* out) {
  HOSTOP_BEGIN(96)
  HOSTOP_TRY(hb.put(0, a, 32));
  if (b) HOSTOP_TRY(hb.put(32, b, 32));
  HOSTOP_TRY(rhip_fr_op(ctx, op, 1, (const rhip_fr*)hb.d, (const rhip_fr*)(hb.d + 32), (rhip_fr*)(hb.d + 64)));
  return hb.get(out, 64, 32);
}
extern "C" int32_t rhip_host_fr_pow(rhip_ctx* ctx, const rhip_fr* a, const rhip_fr* e, rhip_fr* out) {
  return rhip_host_fr_op(ctx, RHIP_FR_POW, a, e, out);
}
extern "C" int32_t rhip_host_g1_on_curve(rhip_ctx* ctx, const rhip_g1* p, int32_t* ok) {
  HOSTOP_BEGIN(128)
  HOSTOP_TRY(hb.put(0, p, 64));
  HOSTOP_TRY(rhip_g1_on_curve(ctx, 1, (const rhip_g1*)hb.d, (uint32_t*)(hb.d + 64)));
  uint32_t v = 0;
  HOSTOP_TRY(hb.get(&v, 64, 4));
  *ok = (int32_t)v;
  return RHIP_OK;
}
extern "C" int32_t rhip_host_g2_on_curve(rhip_ctx* ctx, const rhip_g2* p, int32_t* ok) {
  HOSTOP_BEGIN(256)
  HOSTOP_TRY(hb.put(0, p, 128));
  HOSTOP_TRY(rhip_g2_on_curve(ctx, 1, (const rhip_g2*)hb.d, (uint32_t*)(hb.d + 128)));
  uint32_t v = 0;
  HOSTOP_TRY(hb.get(&v, 128, 4));
  *ok = (int32_t)v;
  return RHIP_OK;
}
// full decoding checks of one element (canonical coordinates + membership): what a `rabe_bn` replacement's deserialisers call
extern "C" int32_t rhip_host_g2_in_subgroup(rhip_ctx* ctx, const rhip_g2* p, int32_t* ok) {
  HOSTOP_BEGIN(256)
  HOSTOP_TRY(hb.put(0, p, 128));
  HOSTOP_TRY(rhip_g2_in_subgroup(ctx, 1, (const rhip_g2*)hb.d, (uint32_t*)(hb.d + 128)));
  uint32_t v = 0;
  HOSTOP_TRY(hb.get(&v, 128, 4));
  *ok = (int32_t)v;
  return RHIP_OK;
}
extern "C" int32_t rhip_host_gt_is_member(rhip_ctx* ctx, const rhip_gt* a, int32_t* ok) {
  HOSTOP_BEGIN(512)
  HOSTOP_TRY(hb.put(0, a, 384));
  HOSTOP_TRY(rhip_gt_is_member(ctx, 1, (const rhip_gt*)hb.d, (uint32_t*)(hb.d + 384)));
  uint32_t v = 0;
  HOSTOP_TRY(hb.get(&v, 384, 4));
  *ok = (int32_t)v;
  return RHIP_OK;
}
extern "C" int32_t rhip_host_fr_from_be32_reduce(rhip_ctx* ctx, const uint8_t digest[32], rhip_fr* out) {
  HOSTOP_BEGIN(64)
  HOSTOP_TRY(hb.put(0, digest, 32));
  HOSTOP_TRY(rhip_fr_from_be32_reduce(ctx, 1, hb.d, (rhip_fr*)(hb.d + 32)));
  return hb.get(out, 32, 32);
}
extern "C" int32_t rhip_host_g1_add(rhip_ctx* ctx, const rhip_g1* a, const rhip_g1* b, rhip_g1* out) {
  HOSTOP_BEGIN(192)
  HOSTOP_TRY(hb.put(0, a, 64)); HOSTOP_TRY(hb.put(64, b, 64));
  HOSTOP_TRY(rhip_g1_add(ctx, 1, (const rhip_g1*)hb.d, (const rhip_g1*)(hb.d + 64), (rhip_g1*)(hb.d + 128)));
  return hb.get(out, 128, 64);
}
extern "C" int32_t rhip_host_g1_neg(rhip_ctx* ctx, const rhip_g1* a, rhip_g1* out) {
  HOSTOP_BEGIN(128)
  HOSTOP_TRY(hb.put(0, a, 64));
  HOSTOP_TRY(rhip_g1_neg(ctx, 1, (const rhip_g1*)hb.d, (rhip_g1*)(hb.d + 64)));
  return hb.get(out, 64, 64);
}
extern "C" int32_t rhip_host_g1_mul(rhip_ctx* ctx, const rhip_g1* p, const rhip_fr* k, rhip_g1* out) {
  HOSTOP_BEGIN(160)
  HOSTOP_TRY(hb.put(0, p, 64)); HOSTOP_TRY(hb.put(64, k, 32));
  HOSTOP_TRY(rhip_g1_mul(ctx, 1, (const rhip_g1*)hb.d, (const rhip_fr*)(hb.d + 64), (rhip_g1*)(hb.d + 96)));
  return hb.get(out, 96, 64);
}
extern "C" int32_t rhip_host_g2_add(rhip_ctx* ctx, const rhip_g2* a, const rhip_g2* b, rhip_g2* out) {
  HOSTOP_BEGIN(384)
  HOSTOP_TRY(hb.put(0, a, 128)); HOSTOP_TRY(hb.put(128, b, 128));
  HOSTOP_TRY(rhip_g2_add(ctx, 1, (const rhip_g2*)hb.d, (const rhip_g2*)(hb.d + 128), (rhip_g2*)(hb.d + 256)));
  return hb.get(out, 256, 128);
}
extern "C" int32_t rhip_host_g2_neg(rhip_ctx* ctx, const rhip_g2* a, rhip_g2* out) {
  HOSTOP_BEGIN(256)
  HOSTOP_TRY(hb.put(0, a, 128));
  HOSTOP_TRY(rhip_g2_neg(ctx, 1, (const rhip_g2*)hb.d, (rhip_g2*)(hb.d + 128)));
  return hb.get(out, 128, 128);
}
extern "C" int32_t rhip_host_g2_mul(rhip_ctx* ctx, const rhip_g2* p, const rhip_fr* k, rhip_g2* out) {
  HOSTOP_BEGIN(288)
  HOSTOP_TRY(hb.put(0, p, 128)); HOSTOP_TRY(hb.put(128, k, 32));
  HOSTOP_TRY(rhip_g2_mul(ctx, 1, (const rhip_g2*)hb.d, (const rhip_fr*)(hb.d + 128), (rhip_g2*)(hb.d + 160)));
  return hb.get(out, 160, 128);
}
extern "C" int32_t rhip_host_gt_mul(rhip_ctx* ctx, const rhip_gt* a, const rhip_gt* b, rhip_gt* out) {
  HOSTOP_BEGIN(1152)
  HOSTOP_TRY(hb.put(0, a, 384)); HOSTOP_TRY(hb.put(384, b, 384));
  HOSTOP_TRY(rhip_gt_mul(ctx, 1, (const rhip_gt*)hb.d, (const rhip_gt*)(hb.d + 384), (rhip_gt*)(hb.d + 768)));
  return hb.get(out, 768, 384);
}
extern "C" int32_t rhip_host_gt_inv(rhip_ctx* ctx, const rhip_gt* a, rhip_gt* out) {
  HOSTOP_BEGIN(768)
  HOSTOP_TRY(hb.put(0, a, 384));
  HOSTOP_TRY(rhip_gt_inv(ctx, 1, (const rhip_gt*)hb.d, (rhip_gt*)(hb.d + 384)));
  return hb.get(out, 384, 384);
}
extern "C" int32_t rhip_host_gt_pow(rhip_ctx* ctx, const rhip_gt* a, const rhip_fr* k, rhip_gt* out) {
  HOSTOP_BEGIN(800)
  HOSTOP_TRY(hb.put(0, a, 384)); HOSTOP_TRY(hb.put(384, k, 32));
  HOSTOP_TRY(rhip_gt_pow(ctx, 1, (const rhip_gt*)hb.d, (const rhip_fr*)(hb.d + 384), (rhip_gt*)(hb.d + 416)));
  return hb.get(out, 416, 384);
}
extern "C" int32_t rhip_host_pairing(rhip_ctx* ctx, const rhip_g1* p, const rhip_g2* q, rhip_gt* out) {
  HOSTOP_BEGIN(576)
  HOSTOP_TRY(hb.put(0, p, 64)); HOSTOP_TRY(hb.put(64, q, 128));
  HOSTOP_TRY(rhip_pairing(ctx, 1, (const rhip_g1*)hb.d, (const rhip_g2*)(hb.d + 64), (rhip_gt*)(hb.d + 192)));
  return hb.get(out, 192, 384);
}

// ------------------------------------------------------------------------------------------------
// tables
template <class TBL, class ENTRY, class BASE, class KERN>
static int32_t table_create(rhip_ctx* ctx, const BASE* host_base, TBL** out, KERN kern, unsigned bs) {
  if (!ctx || !host_base || !out) return RHIP_ERR_ARG;
  *out = nullptr;
  BASE* dbase = nullptr;
  ENTRY* dev = nullptr;
  HIP_TRY(ctx, hipMalloc((void**)&dbase, sizeof(BASE)));
  HIP_TRY(ctx, hipMalloc((void**)&dev, sizeof(ENTRY) * TBL_WINDOWS * TBL_DIGITS));
  HIP_TRY(ctx, hipMemcpyAsync(dbase, host_base, sizeof(BASE), hipMemcpyHostToDevice, ctx->stream));
  KLAUNCH(ctx, "k_table_build", kern, dim3(blocks_for(TBL_WINDOWS * TBL_DIGITS, bs)), dim3(bs), 0, ctx->stream, (const BASE*)dbase, dev);
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipFree(dbase));
  TBL* t = new TBL();
  memset((void*)t, 0, sizeof(TBL));
  t->ctx = ctx;
  t->dev = dev;
  *out = t;
  return RHIP_OK;
}
extern "C" int32_t rhip_g1_table_create(rhip_ctx* ctx, const rhip_g1* b, rhip_g1_table** out) {
  return table_create<rhip_g1_table, G1M>(ctx, b, out, k_table_build_g1, 256);
}
extern "C" int32_t rhip_g2_table_create(rhip_ctx* ctx, const rhip_g2* b, rhip_g2_table** out) {
  return table_create<rhip_g2_table, G2M>(ctx, b, out, k_table_build_g2, 128);
}
extern "C" int32_t rhip_gt_table_create(rhip_ctx* ctx, const rhip_gt* b, rhip_gt_table** out) {
  return table_create<rhip_gt_table, GtM>(ctx, b, out, k_table_build_gt, 64);
}
extern "C" void rhip_g1_table_destroy(rhip_g1_table* t) {
  if (!t) return;
  (void)hipFree(t->dev);
  if (t->dev16) (void)hipFree(t->dev16);
  if (t->wide) (void)hipFree(t->wide);
  delete t;
}
// adds signed w_bits-wide windows (17..27): ceil(254/w) additions per multiplication for ~ 64 B x 2^(w-1) x ceil(254/w)
// of HBM (w = 24: 11 additions, 5.4 GB; w = 26: 10 additions, 19 GB) -- the 288 GB part trades memory for work
extern "C" int32_t rhip_g1_table_add_wide(rhip_ctx* ctx, rhip_g1_table* t, int32_t w_bits) {
  NEED(ctx);
  if (!t || w_bits < 17 || w_bits > 27) return RHIP_ERR_ARG;
  if (t->wide && t->wide_bits == w_bits) return RHIP_OK;
  if (t->wide) { (void)hipFree(t->wide); t->wide = nullptr; t->wide_bits = 0; }
  const int n = wide_windows(w_bits);
  const size_t total = wide_offset(w_bits, n - 1) + wide_count(w_bits, n - 1);
  G1M* d = nullptr;
  HIP_TRY(ctx, hipMalloc((void**)&d, sizeof(G1M) * total));
  for (int i = 0; i < n; i++) {
    const size_t cnt = wide_count(w_bits, i);
    KLAUNCH(ctx, "k_table_build_g1_wide", k_table_build_g1_wide, dim3(blocks_for(cnt, 256)), dim3(256), 0, ctx->stream, (const G1M*)t->dev,
            d + wide_offset(w_bits, i), (int)w_bits, i, cnt);
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  t->wide = d;
  t->wide_bits = w_bits;
  return RHIP_OK;
}
extern "C" int32_t rhip_ac17_pk_set_g_window(rhip_ctx* ctx, rhip_ac17_pk* pk, int32_t w_bits);
// adds the 16-bit-window table (67 MB) to an existing G1 table
extern "C" int32_t rhip_g1_table_add_w16(rhip_ctx* ctx, rhip_g1_table* t) {
  NEED(ctx);
  if (!t) return RHIP_ERR_ARG;
  if (t->dev16) return RHIP_OK;
  G1M* d16 = nullptr;
  const size_t n = (size_t)TBL16_WINDOWS * TBL16_DIGITS;
  HIP_TRY(ctx, hipMalloc((void**)&d16, sizeof(G1M) * n));
  KLAUNCH(ctx, "k_table_build_g1_w16", k_table_build_g1_w16, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream, (const G1M*)t->dev, d16);
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  t->dev16 = d16;
  return RHIP_OK;
}
extern "C" void rhip_g2_table_destroy(rhip_g2_table* t) { if (t) { (void)hipFree(t->dev); if (t->dev16) (void)hipFree(t->dev16); delete t; } }
extern "C" int32_t rhip_g2_table_add_w16(rhip_ctx* ctx, rhip_g2_table* t) {
  NEED(ctx);
  if (!t) return RHIP_ERR_ARG;
  if (t->dev16) return RHIP_OK;
  G2M* d16 = nullptr;
  const size_t n = (size_t)TBL16_WINDOWS * TBL16_DIGITS;
  HIP_TRY(ctx, hipMalloc((void**)&d16, sizeof(G2M) * n));
  KLAUNCH(ctx, "k_table_build_g2_w16", k_table_build_g2_w16, dim3(blocks_for(n, 128)), dim3(128), 0, ctx->stream, (const G2M*)t->dev, d16);
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  t->dev16 = d16;
  return RHIP_OK;
}
extern "C" void rhip_gt_table_destroy(rhip_gt_table* t) { if (t) { (void)hipFree(t->dev); if (t->dev16) (void)hipFree(t->dev16); delete t; } }
extern "C" int32_t rhip_gt_table_add_w16(rhip_ctx* ctx, rhip_gt_table* t) {
  NEED(ctx);
  if (!t) return RHIP_ERR_ARG;
  if (t->dev16) return RHIP_OK;
  GtM* d16 = nullptr;
  const size_t n = (size_t)TBL16_WINDOWS * TBL16_DIGITS;
  HIP_TRY(ctx, hipMalloc((void**)&d16, sizeof(GtM) * n));
  KLAUNCH(ctx, "k_table_build_gt_w16", k_table_build_gt_w16, dim3(blocks_for(n, 64)), dim3(64), 0, ctx->stream, (const GtM*)t->dev, d16);
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  t->dev16 = d16;
  return RHIP_OK;
}
// 16-bit-window tables from 8-bit ones at caller-provided addresses (engine_jobs.hip: AW11's per-attribute tables)
int32_t rhip_build_w16_gt(rhip_ctx* ctx, const GtM* t8, GtM* t16) {
  const size_t n = (size_t)TBL16_WINDOWS * TBL16_DIGITS;
  KLAUNCH(ctx, "k_table_build_gt_w16", k_table_build_gt_w16, dim3(blocks_for(n, 64)), dim3(64), 0, ctx->stream, t8, t16);
  return RHIP_OK;
}
int32_t rhip_build_w16_g2(rhip_ctx* ctx, const G2M* t8, G2M* t16) {
  const size_t n = (size_t)TBL16_WINDOWS * TBL16_DIGITS;
  KLAUNCH(ctx, "k_table_build_g2_w16", k_table_build_g2_w16, dim3(blocks_for(n, 128)), dim3(128), 0, ctx->stream, t8, t16);
  return RHIP_OK;
}
extern "C" int32_t rhip_g1_table_mul(rhip_ctx* ctx, const rhip_g1_table* t, size_t n, const rhip_fr* k, rhip_g1* out) {
  NEED(ctx);
  if (!t) return RHIP_ERR_ARG;
  if (!n) return RHIP_OK;
  KLAUNCH(ctx, "k_table_mul_g1", k_table_mul_g1, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream,
          (const G1M*)(t->dev16 ? t->dev16 : t->dev), n, k, out, t->dev16 ? 1 : 0);
  return RHIP_OK;
}
extern "C" int32_t rhip_g2_table_mul(rhip_ctx* ctx, const rhip_g2_table* t, size_t n, const rhip_fr* k, rhip_g2* out) {
  NEED(ctx);
  if (!t) return RHIP_ERR_ARG;
  if (!n) return RHIP_OK;
  KLAUNCH(ctx, "k_table_mul_g2", k_table_mul_g2, dim3(blocks_for(n, 128)), dim3(128), 0, ctx->stream,
          (const G2M*)(t->dev16 ? t->dev16 : t->dev), n, k, out, t->dev16 ? 1 : 0);
  return RHIP_OK;
}
extern "C" int32_t rhip_gt_table_pow(rhip_ctx* ctx, const rhip_gt_table* t, size_t n, const rhip_fr* k, rhip_gt* out) {
  NEED(ctx);
  if (!t) return RHIP_ERR_ARG;
  if (!n) return RHIP_OK;
  if (rhip_use_c6_gt_pow(ctx, n)) return rhip_launch_gt_table_pow_c6(ctx, t->dev, nullptr, 0, n, k, 1u, nullptr, out);
  KLAUNCH(ctx, "k_table_pow_gt", k_table_pow_gt, dim3(blocks_for(n, 64)), dim3(64), 0, ctx->stream, (const GtM*)t->dev, n, k, out);
  return RHIP_OK;
}

// ------------------------------------------------------------------------------------------------
// AC17 entry points
extern "C" void rhip_ac17_pk_destroy(rhip_ac17_pk* pk) {
  if (!pk) return;
  rhip_g1_table_destroy(pk->g);
  for (int i = 0; i < 3; i++) rhip_g2_table_destroy(pk->h_a[i]);
  for (int i = 0; i < 2; i++) rhip_gt_table_destroy(pk->e[i]);
  delete pk;
}
extern "C" int32_t rhip_ac17_pk_create(rhip_ctx* ctx, const rhip_g1* g, const rhip_g2* h_a, const rhip_gt* e, rhip_ac17_pk** out) {
  if (!ctx || !g || !h_a || !e || !out) return RHIP_ERR_ARG;
  *out = nullptr;
  rhip_ac17_pk* pk = new rhip_ac17_pk();
  pk->ctx = ctx;
  pk->g = nullptr;
  for (int i = 0; i < 3; i++) pk->h_a[i] = nullptr;
  for (int i = 0; i < 2; i++) pk->e[i] = nullptr;
  int32_t rc = rhip_g1_table_create(ctx, g, &pk->g);
  if (!rc) rc = rhip_g1_table_add_w16(ctx, pk->g);
  for (int i = 0; i < 3 && !rc; i++) {
    rc = rhip_g2_table_create(ctx, h_a + i, &pk->h_a[i]);
    if (!rc) rc = rhip_g2_table_add_w16(ctx, pk->h_a[i]);
  }
  for (int i = 0; i < 2 && !rc; i++) {
    rc = rhip_gt_table_create(ctx, e + i, &pk->e[i]);
    if (!rc) rc = rhip_gt_table_add_w16(ctx, pk->e[i]);
  }
  if (rc) { rhip_ac17_pk_destroy(pk); return rc; }
  *out = pk;
  return RHIP_OK;
}
extern "C" int32_t rhip_ac17_pk_set_g_window(rhip_ctx* ctx, rhip_ac17_pk* pk, int32_t w_bits) {
  NEED(ctx);
  if (!pk) return RHIP_ERR_ARG;
  return rhip_g1_table_add_wide(ctx, pk->g, w_bits);
}
static int32_t ac17_enc_c0_launch(rhip_ctx* ctx, const rhip_ac17_pk* pk, size_t n_items, const rhip_fr* s, rhip_g2* c0) {
  const bool g2w16 = pk->h_a[0]->dev16 && pk->h_a[1]->dev16 && pk->h_a[2]->dev16;
  KLAUNCH(ctx, "k_ac17_enc_c0", k_ac17_enc_c0, dim3(blocks_for(n_items * 3, 128)), dim3(128), 0, ctx->stream,
          (const G2M*)(g2w16 ? pk->h_a[0]->dev16 : pk->h_a[0]->dev), (const G2M*)(g2w16 ? pk->h_a[1]->dev16 : pk->h_a[1]->dev),
          (const G2M*)(g2w16 ? pk->h_a[2]->dev16 : pk->h_a[2]->dev), n_items, s, c0, g2w16 ? 1 : 0);
  return RHIP_OK;
}
static int32_t ac17_enc_cp_launch(rhip_ctx* ctx, const rhip_ac17_pk* pk, size_t n_items, const rhip_fr* s, const rhip_gt* msg, rhip_gt* cp) {
  const bool gt16 = pk->e[0]->dev16 && pk->e[1]->dev16;
  if (rhip_use_c6_gt_pow(ctx, n_items))          // small launches: six lanes per running product (engine_coop.hip), the same field elements
    return rhip_launch_gt_table_pow_c6(ctx, gt16 ? pk->e[0]->dev16 : pk->e[0]->dev, gt16 ? pk->e[1]->dev16 : pk->e[1]->dev, gt16 ? 1 : 0, n_items, s, 2u, msg, cp);
  KLAUNCH(ctx, "k_ac17_enc_cp", k_ac17_enc_cp, dim3(blocks_for(n_items, 64)), dim3(64), 0, ctx->stream,
          (const GtM*)(gt16 ? pk->e[0]->dev16 : pk->e[0]->dev), (const GtM*)(gt16 ? pk->e[1]->dev16 : pk->e[1]->dev), n_items, s, msg, cp,
          gt16 ? 1 : 0);
  return RHIP_OK;
}
extern "C" int32_t rhip_ac17_cp_encrypt_batch(rhip_ctx* ctx, const rhip_ac17_pk* pk, size_t n_items, const rhip_fr* A,
                                              const uint32_t* item_A_off, const uint32_t* ct_row_off, size_t total_rows,
                                              const rhip_fr* s, const rhip_gt* msg, rhip_g2* c0, rhip_g1* c, rhip_gt* cp) {
  NEED(ctx);
  if (!pk) return RHIP_ERR_ARG;
  if (!n_items) return RHIP_OK;
  // c (rows), c_0 and c_p of a batch are independent: a launch too small to fill the chip with each of them runs the three side by side
  // (RABE_ENC_FORK=0: one after the other)
  static const int fork_on = getenv("RABE_ENC_FORK") ? atoi(getenv("RABE_ENC_FORK")) : 1;
  const bool forked = fork_on && n_items <= 8192 && total_rows;
  if (forked) { const int32_t rc = rhip_fork(ctx); if (rc) return rc; }
  if (forked) {
    int32_t rc = RHIP_OK;
    {
      RhipOnFork f(ctx, 0);
      rc = ac17_enc_c0_launch(ctx, pk, n_items, s, c0);
    }
    if (!rc) {
      RhipOnFork f(ctx, 1);
      rc = ac17_enc_cp_launch(ctx, pk, n_items, s, msg, cp);
    }
    if (rc) return rc;
  }
  if (total_rows) {
    const rhip_g1_table* g = pk->g;
    KLAUNCH(ctx, "k_ac17_enc_rows", k_ac17_enc_rows, dim3(blocks_for(total_rows, RB_ROWS_BLOCK)), dim3(RB_ROWS_BLOCK), 0, ctx->stream,
            (const G1M*)(g->wide ? g->wide : g->dev16 ? g->dev16 : g->dev), n_items, total_rows, A, item_A_off, ct_row_off, s, c,
            g->wide ? g->wide_bits : g->dev16 ? 1 : 0);
  }
  if (forked) return rhip_join(ctx);
  int32_t rc = ac17_enc_c0_launch(ctx, pk, n_items, s, c0);
  if (rc) return rc;
  return ac17_enc_cp_launch(ctx, pk, n_items, s, msg, cp);
}
extern "C" int32_t rhip_ac17_cp_keygen_batch(rhip_ctx* ctx, const rhip_g1_table* g_table, const rhip_g2_table* h_table, const rhip_g1* g_k,
                                             const rhip_fr* a_inv, const rhip_fr* b, size_t n_items, size_t n_attrs, const rhip_fr* H,
                                             const rhip_fr* H01, const rhip_fr* r, const rhip_fr* sigma, const rhip_fr* sigma_p,
                                             rhip_g2* k0, rhip_g1* k, rhip_g1* kp) {
  NEED(ctx);
  if (!g_table || !h_table) return RHIP_ERR_ARG;
  if (!n_items) return RHIP_OK;
  KLAUNCH(ctx, "k_ac17_keygen_rows", k_ac17_keygen_rows, dim3(blocks_for(n_items * (n_attrs + 1), RB_ROWS_BLOCK)), dim3(RB_ROWS_BLOCK), 0, ctx->stream,
                     (const G1M*)(g_table->wide ? g_table->wide : g_table->dev16 ? g_table->dev16 : g_table->dev), g_k, a_inv, b, n_items, n_attrs, H, H01,
                     r, sigma, sigma_p, k, kp, g_table->wide ? g_table->wide_bits : g_table->dev16 ? 1 : 0);
  KLAUNCH(ctx, "k_ac17_keygen_k0", k_ac17_keygen_k0, dim3(blocks_for(n_items * 3, 128)), dim3(128), 0, ctx->stream,
          (const G2M*)(h_table->dev16 ? h_table->dev16 : h_table->dev), b, n_items, r, k0, h_table->dev16 ? 1 : 0);
  return RHIP_OK;
}
int32_t rhip_ac17_cp_decrypt_batch_lanes6(rhip_ctx* ctx, size_t n_items, const rhip_g2* ct_c0, const rhip_g1* ct_c,
                                              const uint32_t* ct_row_off, const rhip_gt* ct_cp, const rhip_g2* sk_k0, const rhip_g1* sk_k,
                                              const uint32_t* sk_row_off, const rhip_g1* sk_kp, const uint32_t* sk_idx,
                                              const uint32_t* ct_sel, const uint32_t* ct_sel_off, const uint32_t* sk_sel,
                                              const uint32_t* sk_sel_off, rhip_gt* out) {
  NEED(ctx);
  if (!n_items) return RHIP_OK;
  int32_t rc = ensure_scratch(ctx, n_items * 6 * sizeof(GtM));
  if (rc) return rc;
  GtM* mill = (GtM*)ctx->scratch;
  if (rhip_use_c3(ctx, n_items * 6)) {
    KLAUNCH(ctx, "k_ac17_dec_miller_c3", k_ac17_dec_miller_c3, dim3(blocks_for(n_items * 6, C3_TRIPLES_PER_WAVE)), dim3(64), 0, ctx->stream,
            n_items, ct_c0, ct_c, ct_row_off, sk_k0, sk_k, sk_row_off, sk_kp, sk_idx, ct_sel, ct_sel_off, sk_sel, sk_sel_off, mill);
    KLAUNCH(ctx, "k_final_exp_c3", k_final_exp_c3, dim3(blocks_for(n_items, C3_TRIPLES_PER_WAVE)), dim3(64), 0, ctx->stream, n_items,
            (const uint32_t*)nullptr, 6u, (const GtM*)mill, ct_cp, out);
    return RHIP_OK;
  }
  KLAUNCH(ctx, "k_ac17_dec_miller", k_ac17_dec_miller, dim3(blocks_for(n_items * 6, 64)), dim3(64), 0, ctx->stream, n_items, ct_c0, ct_c, ct_row_off,
                     sk_k0, sk_k, sk_row_off, sk_kp, sk_idx, ct_sel, ct_sel_off, sk_sel, sk_sel_off, mill);
  return launch_final_exp(ctx, n_items, (const uint32_t*)nullptr, 6u,
                     (const GtM*)mill, ct_cp, out);
}
extern "C" void rhip_ac17_sk_lines_destroy(rhip_ac17_sk_lines* p);
extern "C" int32_t rhip_ac17_sk_prepare(rhip_ctx* ctx, size_t n_sk, const rhip_g2* sk_k0, rhip_ac17_sk_lines** out) {
  NEED(ctx);
  if (!out || !n_sk || !sk_k0) return RHIP_ERR_ARG;
  rhip_ac17_sk_lines* p = new rhip_ac17_sk_lines{ctx, n_sk, nullptr, nullptr};
  hipError_t e = hipMalloc((void**)&p->lines, n_sk * 3 * RB_MILLER_LINES * sizeof(LineM));
  if (e == hipSuccess) e = hipMalloc((void**)&p->q_inf, n_sk * 3);
  if (e != hipSuccess) {
    if (p->lines) (void)hipFree(p->lines);
    delete p;
    return fail(ctx, e, "rhip_ac17_sk_prepare: hipMalloc");
  }
  KLAUNCH(ctx, "k_g2_prepare_lines", k_g2_prepare_lines, dim3(blocks_for(n_sk * 3, 64)), dim3(64), 0, ctx->stream, n_sk * 3, sk_k0, p->lines, p->q_inf);
  if (rhip_want_lines29(ctx)) { const int32_t rc29 = rhip_lines_to_rr(ctx, n_sk * 3 * RB_MILLER_LINES, p->lines, &p->lines29); if (rc29) { rhip_ac17_sk_lines_destroy(p); return rc29; } }
  // the handle is read by decrypt calls of ANY context (other streams): it must be complete when it is handed out
  hipError_t es = hipStreamSynchronize(ctx->stream);
  if (es != hipSuccess) { rhip_ac17_sk_lines_destroy(p); return fail(ctx, es, "rhip_ac17_sk_prepare: sync"); }
  *out = p;
  return RHIP_OK;
}
// the generic form: lines of n G2 points (bsw: a key's d and d_j.g2; lsw / aw11: whatever side of a batch is fixed)
extern "C" void rhip_g2_lines_destroy(rhip_g2_lines* p) {
  if (!p) return;
  (void)hipFree(p->lines);
  (void)hipFree(p->q_inf);
  if (p->lines29) (void)hipFree(p->lines29);
  delete p;
}
extern "C" int32_t rhip_g2_lines_prepare(rhip_ctx* ctx, size_t n, const rhip_g2* dev_q, rhip_g2_lines** out) {
  NEED(ctx);
  if (!out || !n || !dev_q) return RHIP_ERR_ARG;
  rhip_g2_lines* p = new rhip_g2_lines{ctx, n, nullptr, nullptr};
  hipError_t e = hipMalloc((void**)&p->lines, n * RB_MILLER_LINES * sizeof(LineM));
  if (e == hipSuccess) e = hipMalloc((void**)&p->q_inf, n);
  if (e != hipSuccess) {
    if (p->lines) (void)hipFree(p->lines);
    delete p;
    return fail(ctx, e, "rhip_g2_lines_prepare: hipMalloc");
  }
  KLAUNCH(ctx, "k_g2_prepare_lines", k_g2_prepare_lines, dim3(blocks_for(n, 64)), dim3(64), 0, ctx->stream, n, dev_q, p->lines, p->q_inf);
  if (rhip_want_lines29(ctx)) { const int32_t rc29 = rhip_lines_to_rr(ctx, n * RB_MILLER_LINES, p->lines, &p->lines29); if (rc29) { rhip_g2_lines_destroy(p); return rc29; } }
  hipError_t es = hipStreamSynchronize(ctx->stream);      // the handle is read from any context afterwards
  if (es != hipSuccess) { rhip_g2_lines_destroy(p); return fail(ctx, es, "rhip_g2_lines_prepare: sync"); }
  *out = p;
  return RHIP_OK;
}
extern "C" void rhip_ac17_sk_lines_destroy(rhip_ac17_sk_lines* p) {
  if (!p) return;
  (void)hipFree(p->lines);
  (void)hipFree(p->q_inf);
  if (p->lines29) (void)hipFree(p->lines29);
  delete p;
}
int32_t rhip_ac17_cp_decrypt_batch_prepared_lanes3(rhip_ctx* ctx, size_t n_items, const rhip_g2* ct_c0, const rhip_g1* ct_c,
                                                       const uint32_t* ct_row_off, const rhip_gt* ct_cp, const rhip_ac17_sk_lines* sk_lines,
                                                       const rhip_g1* sk_k, const uint32_t* sk_row_off, const rhip_g1* sk_kp,
                                                       const uint32_t* sk_idx, const uint32_t* ct_sel, const uint32_t* ct_sel_off,
                                                       const uint32_t* sk_sel, const uint32_t* sk_sel_off, rhip_gt* out) {
  NEED(ctx);
  if (!sk_lines) return RHIP_ERR_ARG;
  if (!n_items) return RHIP_OK;
  int32_t rc = ensure_scratch(ctx, n_items * 3 * sizeof(GtM));
  if (rc) return rc;
  GtM* mill = (GtM*)ctx->scratch;
  KLAUNCH(ctx, "k_ac17_dec_miller2", k_ac17_dec_miller2, dim3(blocks_for(n_items * 3, 64)), dim3(64), 0, ctx->stream, n_items, ct_c0, ct_c, ct_row_off,
          (const LineM*)sk_lines->lines, (const uint8_t*)sk_lines->q_inf, sk_k, sk_row_off, sk_kp, sk_idx, ct_sel, ct_sel_off, sk_sel, sk_sel_off,
          mill);
  return launch_final_exp(ctx, n_items, (const uint32_t*)nullptr, 3u,
          (const GtM*)mill, ct_cp, out);
}
