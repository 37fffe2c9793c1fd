// spartan_amd: sparse R1CS matrices on the device (SparseMatPolynomial, src/sparse_mlpoly.rs:19-38).
// F_q has no atomic add, so both products are formulated as gathers over a row-sorted (CSR) and a
// column-sorted (CSC) copy built once at upload (instance setup, outside SNARK::prove).
#include <algorithm>
#include <numeric>

#include "internal.hpp"

struct sp_sparse {
  sp_ctx* ctx;
  size_t nnz, num_rows, num_cols;
  uint32_t *row_ptr, *csr_col, *csr_row;  // CSR (+ row of each entry for evaluate_with_tables)
  Fq* csr_val;
  uint32_t *col_ptr, *csc_row;  // CSC
  Fq* csc_val;
  uint32_t *ent_row, *ent_col;  // the entries in the order they were given (SparseMatPolynomial.M, sparse_mlpoly.rs:19-38): SNARK::encode's
  Fq* ent_val;                  // address lists and value vectors are these, zero-padded (sparse_mlpoly.rs:367-427) — no host pass over them
};

// multiply_vec (sparse_mlpoly.rs:454-464)
__global__ void __launch_bounds__(256) k_spmv(const uint32_t* __restrict__ row_ptr, const uint32_t* __restrict__ col, const Fq* __restrict__ val,
                                              const Fq* __restrict__ z, size_t num_rows, Fq* __restrict__ out) { SP_FG_PRIO();
  size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= num_rows) return;
  Fq acc = fq_zero();
  for (uint32_t e = row_ptr[r]; e < row_ptr[r + 1]; e++) acc = fq_add(acc, fq_mul(ld_fq(val + e), ld_fq(z + col[e])));
  st_fq(out + r, acc);
}
struct Csc3 {
  const uint32_t* col_ptr[3];
  const uint32_t* row[3];
  const Fq* val[3];
  Fq w[3];
  int nm;
};
// compute_eval_table_sparse (sparse_mlpoly.rs:466-481) x nm, combined with weights (r1csproof.rs:275-283)
__global__ void __launch_bounds__(256) k_eval_table(Csc3 M, const Fq* __restrict__ rx, size_t num_cols, Fq* __restrict__ out) { SP_FG_PRIO();
  size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= num_cols) return;
  Fq tot = fq_zero();
  for (int k = 0; k < M.nm; k++) {
    Fq acc = fq_zero();
    for (uint32_t e = M.col_ptr[k][c]; e < M.col_ptr[k][c + 1]; e++) acc = fq_add(acc, fq_mul(ld_fq(rx + M.row[k][e]), ld_fq(M.val[k] + e)));
    tot = fq_add(tot, fq_mul(M.w[k], acc));
  }
  st_fq(out + c, tot);
}
// evaluate_with_tables (sparse_mlpoly.rs:429-438)
__global__ void __launch_bounds__(256) k_sparse_eval(const uint32_t* __restrict__ row, const uint32_t* __restrict__ col, const Fq* __restrict__ val,
                                                     size_t nnz, const Fq* __restrict__ tx, const Fq* __restrict__ ty, Fq* __restrict__ partials) { SP_FG_PRIO();
  __shared__ Fq sm[256];
  Fq acc[1] = {fq_zero()};
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += (size_t)gridDim.x * blockDim.x)
    acc[0] = fq_add(acc[0], fq_mul(fq_mul(ld_fq(tx + row[e]), ld_fq(ty + col[e])), ld_fq(val + e)));
  block_sum_fq<1>(acc, sm);
  if (threadIdx.x == 0) st_fq(partials + blockIdx.x, acc[0]);
}

// several matrices in one launch (grid.y = matrix): partials[m * gridDim.x + blk]
struct SparseMany {
  const uint32_t *row[4], *col[4];
  const Fq* val[4];
  size_t nnz[4];
};
__global__ void __launch_bounds__(256) k_sparse_eval_many(SparseMany M, const Fq* __restrict__ tx, const Fq* __restrict__ ty, Fq* __restrict__ partials) { SP_FG_PRIO();
  __shared__ Fq sm[256];
  const int m = blockIdx.y;
  const uint32_t* __restrict__ row = M.row[m];
  const uint32_t* __restrict__ col = M.col[m];
  const Fq* __restrict__ val = M.val[m];
  Fq acc[1] = {fq_zero()};
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < M.nnz[m]; e += (size_t)gridDim.x * blockDim.x)
    acc[0] = fq_add(acc[0], fq_mul(fq_mul(ld_fq(tx + row[e]), ld_fq(ty + col[e])), ld_fq(val + e)));
  block_sum_fq<1>(acc, sm);
  if (threadIdx.x == 0) st_fq(partials + (size_t)m * gridDim.x + blockIdx.x, acc[0]);
}
__global__ void __launch_bounds__(256) k_sparse_sums(const Fq* __restrict__ partials, size_t nblk, Fq* __restrict__ out) { SP_FG_PRIO();  // block per matrix
  __shared__ Fq sm[256];
  Fq acc[1] = {fq_zero()};
  for (size_t b = threadIdx.x; b < nblk; b += 256) acc[0] = fq_add(acc[0], ld_fq(partials + (size_t)blockIdx.x * nblk + b));
  block_sum_fq<1>(acc, sm);
  if (threadIdx.x == 0) st_fq(out + blockIdx.x, acc[0]);
}

template <typename T>
static int32_t dev_put(sp_ctx* c, T** d, const std::vector<T>& h) {
  size_t bytes = sizeof(T) * (h.size() ? h.size() : 1);
  HIPCHK(hipMalloc((void**)d, bytes));
  if (!h.empty()) HIPCHK(hipMemcpy(*d, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice));
  return SP_OK;
}

extern "C" {

int32_t sp_sparse_upload(sp_ctx* c, const uint64_t* rows, const uint64_t* cols, const uint64_t* vals, size_t nnz, size_t num_rows,
                         size_t num_cols, sp_sparse** out) {
  if (!c || !out || (nnz && (!rows || !cols || !vals)) || num_rows == 0 || num_cols == 0 || nnz >= 0xffffffffu) return SP_EINVAL;
  for (size_t i = 0; i < nnz; i++)
    if (rows[i] >= num_rows || cols[i] >= num_cols) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  sp_sparse* m = new (std::nothrow) sp_sparse();
  if (!m) return SP_ENOMEM;
  memset(m, 0, sizeof *m);
  m->ctx = c; m->nnz = nnz; m->num_rows = num_rows; m->num_cols = num_cols;
  const Fq* v = (const Fq*)vals;
  auto build = [&](const uint64_t* key, size_t nkeys, const uint64_t* other, std::vector<uint32_t>& ptr, std::vector<uint32_t>& oth,
                   std::vector<uint32_t>& keys_sorted, std::vector<Fq>& vv) {
    ptr.assign(nkeys + 1, 0);
    for (size_t i = 0; i < nnz; i++) ptr[key[i] + 1]++;
    for (size_t k = 0; k < nkeys; k++) ptr[k + 1] += ptr[k];
    std::vector<uint32_t> cur(ptr.begin(), ptr.end() - 1);
    oth.resize(nnz); vv.resize(nnz); keys_sorted.resize(nnz);
    for (size_t i = 0; i < nnz; i++) {  // stable: preserves the entry order within a key
      uint32_t pos = cur[key[i]]++;
      oth[pos] = (uint32_t)other[i];
      keys_sorted[pos] = (uint32_t)key[i];
      vv[pos] = v[i];
    }
  };
  std::vector<uint32_t> ptr, oth, ks;
  std::vector<Fq> vv;
  int32_t rc;
  build(rows, num_rows, cols, ptr, oth, ks, vv);
  if ((rc = dev_put(c, &m->row_ptr, ptr)) || (rc = dev_put(c, &m->csr_col, oth)) || (rc = dev_put(c, &m->csr_row, ks)) || (rc = dev_put(c, &m->csr_val, vv))) { sp_sparse_free(m); return rc; }
  build(cols, num_cols, rows, ptr, oth, ks, vv);
  if ((rc = dev_put(c, &m->col_ptr, ptr)) || (rc = dev_put(c, &m->csc_row, oth)) || (rc = dev_put(c, &m->csc_val, vv))) { sp_sparse_free(m); return rc; }
  {
    std::vector<uint32_t> er(nnz), ec(nnz);
    for (size_t i = 0; i < nnz; i++) { er[i] = (uint32_t)rows[i]; ec[i] = (uint32_t)cols[i]; }
    std::vector<Fq> ev(v, v + nnz);
    if ((rc = dev_put(c, &m->ent_row, er)) || (rc = dev_put(c, &m->ent_col, ec)) || (rc = dev_put(c, &m->ent_val, ev))) { sp_sparse_free(m); return rc; }
  }
  *out = m;
  return SP_OK;
}
// SNARK::encode (lib.rs:325-336 -> sparse_mlpoly.rs:367-427): the row / column addresses of a matrix's entries as a device index list of
// n >= nnz elements (zero-padded: MultiSparseMatPolynomialAsDense pads every matrix to the batch's num_nz_entries), and its values written
// into dst[dst_off, dst_off + n) — both from the entry-order copies kept at upload, device to device.
int32_t sp_sparse_entry_index(sp_ctx* c, const sp_sparse* m, int which, size_t n, sp_index** out) {
  if (!c || !m || !out || (which != 0 && which != 1) || n == 0 || n < m->nnz) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  sp_index* ix = new (std::nothrow) sp_index();
  if (!ix) return SP_ENOMEM;
  ix->ctx = c; ix->n = n; ix->d = nullptr;
  hipError_t e = hipMalloc((void**)&ix->d, 4 * n);
  if (e == hipSuccess && n > m->nnz) e = hipMemsetAsync(ix->d + m->nnz, 0, 4 * (n - m->nnz), c->stream);
  if (e == hipSuccess && m->nnz) e = hipMemcpyAsync(ix->d, which == 0 ? m->ent_row : m->ent_col, 4 * m->nnz, hipMemcpyDeviceToDevice, c->stream);
  if (e != hipSuccess) { if (ix->d) (void)hipFree(ix->d); delete ix; return e == hipErrorOutOfMemory ? SP_ENOMEM : SP_EHIP; }
  *out = ix;
  return SP_OK;
}
int32_t sp_sparse_entry_values(sp_ctx* c, const sp_sparse* m, sp_table* dst, size_t dst_off, size_t n) {
  if (!c || !m || !dst || n < m->nnz || dst_off + n > dst->cap) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  if (n > m->nnz) HIPCHK(hipMemsetAsync(dst->d + dst_off + m->nnz, 0, 32 * (n - m->nnz), c->stream));
  if (m->nnz) HIPCHK(hipMemcpyAsync(dst->d + dst_off, m->ent_val, 32 * m->nnz, hipMemcpyDeviceToDevice, c->stream));
  return SP_OK;
}
void sp_sparse_free(sp_sparse* m) {
  if (!m) return;
  (void)hipSetDevice(m->ctx->dev);
  (void)hipStreamSynchronize(m->ctx->stream);
  void* ps[] = {m->row_ptr, m->csr_col, m->csr_row, m->csr_val, m->col_ptr, m->csc_row, m->csc_val, m->ent_row, m->ent_col, m->ent_val};
  for (void* p : ps)
    if (p) (void)hipFree(p);
  delete m;
}
int32_t sp_sparse_mulvec(sp_ctx* c, const sp_sparse* m, const sp_table* z, sp_table** out) {
  if (!c || !m || !z || !out || z->len < m->num_cols) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  SPCHK(table_new(c, m->num_rows, false, out));
  {
    ProfScope ps(c, PF_SPARSE, (double)m->nnz * (4 + 32 + 32) + 32.0 * (double)m->num_rows);
    hipLaunchKernelGGL(k_spmv, dim3((unsigned)((m->num_rows + 255) / 256)), dim3(256), 0, c->stream, (const uint32_t*)m->row_ptr,
                       (const uint32_t*)m->csr_col, (const Fq*)m->csr_val, (const Fq*)z->d, m->num_rows, (*out)->d);
  }
  return hipGetLastError() == hipSuccess ? SP_OK : SP_EHIP;
}
int32_t sp_sparse_eval_table(sp_ctx* c, const sp_sparse* const* ms, const uint64_t* w, size_t nm, const sp_table* rx, sp_table** out) {
  if (!c || !ms || !w || !rx || !out || nm == 0 || nm > 3) return SP_EINVAL;
  Csc3 M;
  memset(&M, 0, sizeof M);
  M.nm = (int)nm;
  double bytes = 0;
  for (size_t k = 0; k < nm; k++) {
    if (!ms[k] || ms[k]->num_cols != ms[0]->num_cols || rx->len < ms[k]->num_rows) return SP_EINVAL;
    M.col_ptr[k] = ms[k]->col_ptr; M.row[k] = ms[k]->csc_row; M.val[k] = ms[k]->csc_val;
    memcpy(M.w[k].l, w + 4 * k, 32);
    bytes += (double)ms[k]->nnz * (4 + 32 + 32);
  }
  HIPCHK(hipSetDevice(c->dev));
  size_t nc = ms[0]->num_cols;
  SPCHK(table_new(c, nc, false, out));
  {
    ProfScope ps(c, PF_SPARSE, bytes + 32.0 * (double)nc);
    hipLaunchKernelGGL(k_eval_table, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, c->stream, M, (const Fq*)rx->d, nc, (*out)->d);
  }
  return hipGetLastError() == hipSuccess ? SP_OK : SP_EHIP;
}
int32_t sp_sparse_evaluate(sp_ctx* c, const sp_sparse* m, const sp_table* tx, const sp_table* ty, uint64_t out[4]) {
  if (!c || !m || !tx || !ty || !out || tx->len < m->num_rows || ty->len < m->num_cols) return SP_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  size_t nblk = grid_for(m->nnz ? m->nnz : 1, 1024);
  SPCHK(ensure(&c->scratch, &c->scratch_cap, 32 * (nblk + 1)));
  Fq* partials = partials_dst(c, nblk, 1);
  {
    ProfScope ps(c, PF_SPARSE, (double)m->nnz * (8 + 96));
    hipLaunchKernelGGL(k_sparse_eval, dim3((unsigned)nblk), dim3(256), 0, c->stream, (const uint32_t*)m->csr_row, (const uint32_t*)m->csr_col,
                       (const Fq*)m->csr_val, m->nnz, (const Fq*)tx->d, (const Fq*)ty->d, partials);
  }
  return reduce_and_fetch(c, partials, nblk, 1, out);
}

// R1CSInstance::evaluate (r1cs.rs:300-303) of up to four matrices, queued on a low-priority stream of its own and collected with
// sp_job_wait (32 bytes per matrix): SNARK::prove starts it as soon as the second sum-check has fixed ry, and the kernels fill the
// idle time of the witness opening (an inner-product argument: launch-sized kernels and host trips) instead of standing on the
// critical path behind it. tx, ty must stay alive until the job has been waited for.
int32_t sp_sparse_evaluate_begin(sp_ctx* c, const sp_sparse* const* ms, size_t count, const sp_table* tx, const sp_table* ty, sp_job** out) {
  if (!c || !ms || !tx || !ty || !out || count == 0 || count > 4) return SP_EINVAL;
  SparseMany M;
  memset(&M, 0, sizeof M);
  size_t max_nnz = 1;
  double bytes = 0;
  for (size_t k = 0; k < count; k++) {
    if (!ms[k] || tx->len < ms[k]->num_rows || ty->len < ms[k]->num_cols) return SP_EINVAL;
    M.row[k] = ms[k]->csr_row; M.col[k] = ms[k]->csr_col; M.val[k] = ms[k]->csr_val; M.nnz[k] = ms[k]->nnz;
    if (ms[k]->nnz > max_nnz) max_nnz = ms[k]->nnz;
    bytes += (double)ms[k]->nnz * (8 + 96);
  }
  HIPCHK(hipSetDevice(c->dev));
  if (!c->stream_low) return SP_EHIP;
  const size_t nblk = grid_for(max_nnz, 1024);
  sp_job* j = new (std::nothrow) sp_job();
  if (!j) return SP_ENOMEM;
  j->ctx = c; j->rows = count; j->stream = c->stream_low; j->scratch = nullptr;
  j->out_off = 32 * nblk * count;
  j->scratch_bytes = j->out_off + 32 * count;
  int32_t rc = pool_alloc(c, j->scratch_bytes, (void**)&j->scratch);
  if (rc != SP_OK) { delete j; return rc; }
  hipEvent_t ready;
  if (hipEventCreateWithFlags(&ready, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&j->done, hipEventDisableTiming) != hipSuccess) {
    pool_release(c, j->scratch, j->scratch_bytes); delete j; return SP_EHIP;
  }
  (void)hipEventRecord(ready, c->stream);  // tx, ty as produced by what is queued on the main stream
  (void)hipStreamWaitEvent(c->stream_low, ready, 0);
  {
    ProfScope ps(c, PF_SPARSE, bytes, c->stream_low);
    hipLaunchKernelGGL(k_sparse_eval_many, dim3((unsigned)nblk, (unsigned)count), dim3(256), 0, c->stream_low, M, (const Fq*)tx->d, (const Fq*)ty->d, (Fq*)j->scratch);
    hipLaunchKernelGGL(k_sparse_sums, dim3((unsigned)count), dim3(256), 0, c->stream_low, (const Fq*)j->scratch, nblk, (Fq*)(j->scratch + j->out_off));
  }
  (void)hipEventRecord(j->done, c->stream_low);
  (void)hipEventDestroy(ready);
  if (hipGetLastError() != hipSuccess) { (void)hipEventDestroy(j->done); pool_release(c, j->scratch, j->scratch_bytes); delete j; return SP_EHIP; }
  *out = j;
  return SP_OK;
}

}  // extern "C"
