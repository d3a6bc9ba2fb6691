// m3hip_par.hip -- the parameter handler's O(nPar^2) work of a Metropolis step on the device (m3hip_par_*, include/m3hip.h).
#include "m3hip_ctx.h"

void par_free(m3hip_handle h, m3hip_ctx::ParDev& p) {  // buffers only: the slot stays live
  if (h->par_stream) (void)hipStreamSynchronize(h->par_stream);
  else if (h->stream) (void)hipStreamSynchronize(h->stream);
  dev_free(p.d_inv); dev_free(p.d_chol); dev_free(p.d_rows); dev_free(p.d_in);
  for (int i = 0; i < 2; ++i)
    if (p.h_in[i]) (void)hipHostFree(p.h_in[i]);
  if (p.h_out) (void)hipHostFree(p.h_out);
  const bool live = p.live;
  p = m3hip_ctx::ParDev();
  p.live = live;
}

extern "C" {

/* ---------------------------------------------------------------- parameter handler on the device */

static m3hip_ctx::ParDev* par_slot(m3hip_handle h, int slot, const char* who) {
  if (slot < 0 || slot >= int(h->pars.size()) || !h->pars[size_t(slot)].live) {
    fail(h, M3HIP_EINVAL, std::string(who) + ": no such parameter-handler slot (m3hip_par_create)");
    return nullptr;
  }
  return &h->pars[size_t(slot)];
}

static int par_wait(m3hip_handle h, m3hip_ctx::ParDev& p) {  // every launched step of this slot has published its results
  if (p.collected == p.ticket) return M3HIP_OK;
  volatile uint64_t* flag = reinterpret_cast<volatile uint64_t*>(p.h_out + 1 + p.n);
  hipStream_t st = h->par_stream ? h->par_stream : h->stream;
  const double t0 = now_s();
  uint64_t spins = 0;
  (void)hipStreamQuery(st);
  while (*flag != p.ticket) {
    if ((++spins & 0xFFu) == 0) {
      hipError_t e = hipStreamQuery(st);
      if (e != hipSuccess && e != hipErrorNotReady) return fail(h, M3HIP_EHIP, std::string("par_result: ") + hipGetErrorString(e));
      // the stream has drained and the ticket is not there: the kernels that would publish it never ran
      if (e == hipSuccess && *flag != p.ticket && now_s() - t0 > 2.0) return fail(h, M3HIP_EHIP, "par_result: the stream is idle but the step's results were never published");
      if (now_s() - t0 > 60.0) return fail(h, M3HIP_EHIP, "par_result: timed out waiting for the device");
    }
  }
  (void)hipStreamQuery(st);
  p.collected = p.ticket;
  return M3HIP_OK;
}

int m3hip_par_create(m3hip_handle h, int* slot_out) try {
  if (!h || !slot_out) return M3HIP_EINVAL;
  for (size_t i = 0; i < h->pars.size(); ++i)
    if (!h->pars[i].live) {
      h->pars[i] = m3hip_ctx::ParDev();
      h->pars[i].live = true;
      *slot_out = int(i);
      return M3HIP_OK;
    }
  if (h->pars.size() >= size_t(M3HIP_MAX_PAR_SLOTS)) return fail(h, M3HIP_ESTATE, "par_create: all " + std::to_string(M3HIP_MAX_PAR_SLOTS) + " parameter-handler slots are in use");
  h->pars.emplace_back();
  h->pars.back().live = true;
  *slot_out = int(h->pars.size()) - 1;
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_par_destroy(m3hip_handle h, int slot) try {
  if (!h) return M3HIP_EINVAL;
  m3hip_ctx::ParDev* p = par_slot(h, slot, "par_destroy");
  if (!p) return M3HIP_EINVAL;
  (void)hipSetDevice(h->device);
  par_free(h, *p);
  p->live = false;
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_par_upload(m3hip_handle h, int slot, int which, const double* m, int n) try {
  if (!h) return M3HIP_EINVAL;
  m3hip_ctx::ParDev* pp = par_slot(h, slot, "par_upload");
  if (!pp) return M3HIP_EINVAL;
  if ((which != 0 && which != 1) || !m || n < 1 || n > 16384) return fail(h, M3HIP_EINVAL, "par_upload: which is 0 (inverse covariance) or 1 (Cholesky factor), 1 <= n <= 16384");
  HIPCHK(h, hipSetDevice(h->device));
  auto& p = *pp;
  int rc;
  if (p.n != n) {  // (re)size everything
    par_free(h, p);
    if (!h->par_stream) HIPCHK(h, hipStreamCreateWithFlags(&h->par_stream, hipStreamNonBlocking));
    if ((rc = dev_alloc(h, &p.d_inv, size_t(n) * n))) return rc;
    if ((rc = dev_alloc(h, &p.d_chol, size_t(n) * n))) return rc;
    if ((rc = dev_alloc(h, &p.d_rows, size_t(2) * n))) return rc;
    if ((rc = dev_alloc(h, &p.d_in, size_t(2) * n))) return rc;
    for (int i = 0; i < 2; ++i) {
      HIPCHK(h, hipHostMalloc((void**)&p.h_in[i], sizeof(double) * 2 * n, hipHostMallocMapped));
      HIPCHK(h, hipHostGetDevicePointer((void**)&p.d_in_host[i], p.h_in[i], 0));
    }
    HIPCHK(h, hipHostMalloc((void**)&p.h_out, sizeof(double) * (n + 2), hipHostMallocMapped));
    memset(p.h_out, 0, sizeof(double) * (n + 2));
    HIPCHK(h, hipHostGetDevicePointer((void**)&p.d_out, p.h_out, 0));
    p.n = n;
  }
  if ((rc = par_wait(h, p))) return rc;  // a step in flight still reads the old matrix
  HIPCHK(h, hipMemcpy(which == 0 ? p.d_inv : p.d_chol, m, sizeof(double) * size_t(n) * n, hipMemcpyHostToDevice));
  (which == 0 ? p.have_inv : p.have_chol) = true;
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_par_launch(m3hip_handle h, int slot, const double* d, const double* z, int n) try {
  if (!h) return M3HIP_EINVAL;
  m3hip_ctx::ParDev* pp = par_slot(h, slot, "par_launch");
  if (!pp) return M3HIP_EINVAL;
  auto& p = *pp;
  if (!p.have_inv || !p.have_chol) return fail(h, M3HIP_ESTATE, "par_launch before both matrices were uploaded (m3hip_par_upload)");
  if (!d || !z || n != p.n) return fail(h, M3HIP_EINVAL, "par_launch: NULL vectors or n differs from the uploaded matrices");
  HIPCHK(h, hipSetDevice(h->device));
  int rc;
  if (p.ticket > p.collected && (rc = par_wait(h, p))) return rc;  // results nobody collected: the output block is about to be rewritten
  const int in = int(p.ticket & 1);
  memcpy(p.h_in[in], d, sizeof(double) * n);
  memcpy(p.h_in[in] + n, z, sizeof(double) * n);
  // the ticket counts a step only once its launches went out: a failed launch must not leave every later wait spinning
  // for a ticket no kernel will ever publish
  const uint64_t ticket = p.ticket + 1;
  launch_par_step(p.d_inv, p.d_chol, p.d_in_host[in], p.d_in, n, p.d_rows, p.d_out, reinterpret_cast<volatile uint64_t*>(p.d_out + 1 + n), ticket,
                  h->par_stream ? h->par_stream : h->stream);
  HIPCHK(h, hipGetLastError());
  p.ticket = ticket;
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_par_result(m3hip_handle h, int slot, double* prior, double* corr_throw, int n) try {
  if (!h) return M3HIP_EINVAL;
  m3hip_ctx::ParDev* pp = par_slot(h, slot, "par_result");
  if (!pp) return M3HIP_EINVAL;
  auto& p = *pp;
  if (p.ticket == 0) return fail(h, M3HIP_ESTATE, "par_result before par_launch");
  if (n != p.n) return fail(h, M3HIP_EINVAL, "par_result: n differs from the uploaded matrices");
  int rc = par_wait(h, p);
  if (rc) return rc;
  if (prior) *prior = p.h_out[0];
  if (corr_throw) memcpy(corr_throw, p.h_out + 1, sizeof(double) * n);
  return M3HIP_OK;
} M3_API_CATCH(h)

}  // extern "C"
