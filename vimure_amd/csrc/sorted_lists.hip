// sorted_lists.hip -- builds the sorted report lists (layout: sweep_sl.h) from tie-major entries, once per dataset, and
// the permutation kernels of the boundary functions.  Replaces the data set-up of `__check_fit_params` (model.py:136-171)
// together with vmr_create / vmr_create_coo in vimure_hip.hip.
#include "sweep_sl.h"

// keys = per-tie report counts, values = tie index (one layer)
__global__ __launch_bounds__(256) void k_sl_keys(const unsigned* __restrict__ cnt, unsigned* __restrict__ keys, unsigned* __restrict__ vals, size_t T) {
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < T; t += (size_t)gridDim.x * 256) { keys[t] = cnt[t]; vals[t] = (unsigned)t; }
}
// Second sort key of layers with long lists (most steps then run the sweep's general body): among the ties of equal report count,
// those with more reports of mirror count >= 1 first -- keys[t] = count << nb | n1.  The ties of a step then agree on n1 too, and
// with every tie's count->=-1 reports first in its list (k_far_first) the last count - n1 rounds of a step hold mirror count 0
// only: the statistics pass adds nothing to its LDS table there (SlArgs::h0s).  rpl: scanned counts; E: the layer's entries, tie-major.
__global__ __launch_bounds__(256) void k_sl_level_keys(unsigned* __restrict__ keys, const unsigned* __restrict__ rpl, const unsigned* __restrict__ E,
                                                       size_t T, unsigned rows0, int nb) {
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < T; t += (size_t)gridDim.x * 256) {
    const unsigned r0 = rpl[t], n = rpl[t + 1] - r0;
    unsigned n1 = 0;
    for (unsigned r = 0; r < n; ++r) n1 += SL_YM(E[(size_t)r0 + r]) >= rows0 ? 1u : 0u;
    keys[t] = (n << nb) | n1;
  }
}
// per step: slots = 64 * (count of the step's first = most reported tie); the sorted order padded to whole steps
// (nb: low bits of a key that hold the second sort key, k_sl_level_keys)
__global__ __launch_bounds__(256) void k_sl_steps(const unsigned* __restrict__ keys_sorted, const unsigned* __restrict__ vals_sorted,
                                                  unsigned* __restrict__ sz, unsigned* __restrict__ perm, size_t T, size_t NS, int nb) {
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < NS * 64; q += (size_t)gridDim.x * 256) {
    perm[q] = q < T ? vals_sorted[q] : 0xffffffffu;
    if ((q & 63) == 0) sz[q >> 6] = 64u * (keys_sorted[q] >> nb);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) sz[NS] = 0u;
}
// one wave per step: lane <-> position; round r of the step = the r-th entry of every tie (0 where it has fewer)
// (wide entries, Geo::wide: Ein2 / Eout2 carry the second word of every entry through the same placement; the first word is then
// the table row itself)
__global__ __launch_bounds__(256) void k_sl_place(const unsigned* __restrict__ perm, const unsigned* __restrict__ rpl /*scanned, by tie*/,
                                                  const unsigned* __restrict__ rsl, const unsigned* __restrict__ Ein,
                                                  unsigned* __restrict__ Eout, const unsigned* __restrict__ Ein2, unsigned* __restrict__ Eout2,
                                                  unsigned* __restrict__ syl, size_t T, size_t NS, int Mp) {
  const int lane = threadIdx.x & 63;
  for (size_t s = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); s < NS; s += (size_t)gridDim.x * 4) {
    const unsigned t = perm[s * 64 + lane];
    const bool ok = t != 0xffffffffu;
    const unsigned r0 = ok ? rpl[t] : 0u, n = ok ? rpl[t + 1] - r0 : 0u;
    const unsigned ea = rsl[s], R = (rsl[s + 1] - ea) >> 6;
    // A tie's reports arrive in reporter order.  Left like that, round r would hold the r-th smallest reporter of all 64
    // ties -- a narrow band of table rows, several lanes on the SAME row: the LDS float atomics of walk 2 then serialise
    // (config-5 layer: 77 cycles per wave-instruction against ~25).  Every tie's list is therefore rotated by a per-tie
    // pseudo-random offset, which spreads a round over the whole reporter range.
    unsigned rot = 0;
    if (n > 1) { unsigned hsh = t * 0x9E3779B1u; hsh ^= hsh >> 15; hsh *= 0x85EBCA77u; hsh ^= hsh >> 13; rot = hsh % n; }
    unsigned ymx = 0;
    for (unsigned r = 0; r < R; ++r) {
      unsigned q = r + rot;
      if (q >= n) q -= n;
      const unsigned e = n > r ? Ein[(size_t)r0 + q] : 0u;
      Eout[(size_t)ea + r * 64 + lane] = e;
      if (Ein2) Eout2[(size_t)ea + r * 64 + lane] = n > r ? Ein2[(size_t)r0 + q] : 0u;
      ymx = max(ymx, (Ein2 ? e : SL_YM(e)) / (unsigned)Mp);
    }
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) ymx = max(ymx, (unsigned)__shfl_xor((int)ymx, o2, 64));
    if (lane == 0) syl[s] = ymx;
  }
}

template <class V>
__global__ __launch_bounds__(256) void k_sl_gather(const unsigned* __restrict__ perm, const V* __restrict__ in, V* __restrict__ out,
                                                   size_t T, size_t NS, int L) {
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < (size_t)L * T; q += (size_t)gridDim.x * 256) {
    const size_t l = q / T, pos = q - l * T;
    out[q] = in[l * T + perm[l * NS * 64 + pos]];
  }
}
// rows of K doubles: out[pos] = in[perm[pos]] (to_pos) or out[perm[pos]] = in[pos]; K threads per row
__global__ __launch_bounds__(256) void k_sl_rows(const unsigned* __restrict__ perm, const double* __restrict__ in, double* __restrict__ out,
                                                 size_t T, size_t NS, int L, int K, int to_pos) {
  const size_t n = (size_t)L * T * K;
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (size_t)gridDim.x * 256) {
    const size_t row = q / K, k = q - row * K, l = row / T, pos = row - l * T;
    const size_t t = perm[l * NS * 64 + pos];
    if (to_pos) out[q] = in[(l * T + t) * K + k]; else out[(l * T + t) * K + k] = in[q];
  }
}

int sl_permute_u8(vmr_ctx* h, const uint8_t* in, uint8_t* out) {
  const Geo& g = h->g;
  const size_t T = (size_t)g.N * g.N, NS = (T + 63) / 64;
  hipLaunchKernelGGL(k_sl_gather<uint8_t>, dim3((unsigned)std::min<size_t>(8192, (g.L * T + 255) / 256)), dim3(256), 0, h->stream, h->perm, in, out, T, NS, g.L);
  CK(hipGetLastError());
  return VMR_OK;
}
int sl_permute_u32(vmr_ctx* h, const unsigned* in, unsigned* out) {
  const Geo& g = h->g;
  const size_t T = (size_t)g.N * g.N, NS = (T + 63) / 64;
  hipLaunchKernelGGL(k_sl_gather<unsigned>, dim3((unsigned)std::min<size_t>(8192, (g.L * T + 255) / 256)), dim3(256), 0, h->stream, h->perm, in, out, T, NS, g.L);
  CK(hipGetLastError());
  return VMR_OK;
}
int sl_permute_rows(vmr_ctx* h, const double* in, double* out, bool to_pos) {
  const Geo& g = h->g;
  const size_t T = (size_t)g.N * g.N, NS = (T + 63) / 64, n = (size_t)g.L * T * g.K;
  hipLaunchKernelGGL(k_sl_rows, dim3((unsigned)std::min<size_t>(16384, (n + 255) / 256)), dim3(256), 0, h->stream, h->perm, in, out, T, NS, g.L, g.K, to_pos ? 1 : 0);
  HIPCHK(h, hipGetLastError());
  return VMR_OK;
}

int sl_place_entries(vmr_ctx* h, unsigned* rp, const std::vector<unsigned long long>& nl, unsigned* etmp_all, unsigned* etmp2_all, const SlFill* fill) {
  Geo& g = h->g;
  const int L = g.L;
  const size_t T = (size_t)g.N * g.N, n = T + 1, NS = (T + 63) / 64;
  unsigned *keys = nullptr, *keys2 = nullptr, *vals = nullptr, *vals2 = nullptr, *bsum = nullptr, *etmp = nullptr, *etmp2 = nullptr;
  void* tmp = nullptr;
  const bool wide = g.wide != 0;
  auto cleanup = [&]() { void* p[] = {keys, keys2, vals, vals2, bsum, etmp, etmp2, tmp}; for (void* q : p) if (q) (void)hipFree(q); };
#define CKS(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { g_create_err = std::string(#call) + ": " + hipGetErrorString(e_); (void)hipGetLastError(); cleanup(); return VMR_EHIP; } } while (0)
  if (T >= 0x7fffffffull) return fail(nullptr, VMR_EINVAL, "more than 2^31 ties in one layer");
  CKS(hipMalloc(&keys, T * 4)); CKS(hipMalloc(&keys2, T * 4)); CKS(hipMalloc(&vals, T * 4)); CKS(hipMalloc(&vals2, T * 4));
  CKS(hipMalloc(&bsum, (std::max(n, NS + 1) + 2047) / 2048 * 4));
  CKS(hipMalloc(&h->rs, (size_t)L * (NS + 1) * 4));
  CKS(hipMalloc(&h->perm, (size_t)L * NS * 64 * 4));
  CKS(hipMalloc(&h->sy, (size_t)L * NS * 4));
  size_t tb = 0;
  int bits = 1;
  while (bits < 32 && (1ull << bits) <= (unsigned long long)g.M) ++bits;   // a tie holds at most M reports
  // (layers of long lists: a second key, k_sl_level_keys -- packed entries of a mutual network only)
  const bool level_keys_ok = !wide && g.mut && 2 * bits <= 30 && !getenv("VMR_NO_LEVEL_SORT");
  CKS(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, tb, keys, keys2, vals, vals2, (int)T, 0, level_keys_ok ? 2 * bits : bits, h->stream));
  CKS(hipMalloc(&tmp, tb ? tb : 8));
  const unsigned tgrid = (unsigned)std::min<size_t>(8192, (T + 255) / 256), sgrid = (unsigned)std::min<size_t>(8192, (NS + 3) / 4);
  if (!etmp_all) {
    unsigned long long nlmax = 0;
    for (int l = 0; l < L; ++l) nlmax = std::max(nlmax, nl[l]);
    CKS(hipMalloc(&etmp, ((size_t)nlmax + 64) * 4));
    if (wide) CKS(hipMalloc(&etmp2, ((size_t)nlmax + 64) * 4));
  }
  unsigned long long off1 = 0;
  for (int l = 0; l < L; ++l) {
    unsigned* rpl = rp + (size_t)l * n;
    unsigned* rsl = h->rs + (size_t)l * (NS + 1);
    const bool by_level = level_keys_ok && (double)nl[l] > (double)SL_PF * (double)T;
    const int nb = by_level ? bits : 0;
    int rc;
    hipLaunchKernelGGL(k_sl_keys, dim3(tgrid), dim3(256), 0, h->stream, rpl, keys, vals, T);
    if (by_level) {
      if ((rc = scan_u32(h, rpl, bsum, n))) { cleanup(); return rc; }
      if (!etmp_all) (*fill)(l, rpl, etmp, etmp2);
      hipLaunchKernelGGL(k_sl_level_keys, dim3(tgrid), dim3(256), 0, h->stream, keys, rpl, etmp_all ? etmp_all + off1 : etmp, T, (unsigned)g.Mp, nb);
    }
    CKS(hipcub::DeviceRadixSort::SortPairsDescending(tmp, tb, keys, keys2, vals, vals2, (int)T, 0, bits + nb, h->stream));   // (stable: equal keys keep tie order)
    hipLaunchKernelGGL(k_sl_steps, dim3((unsigned)std::min<size_t>(8192, (NS * 64 + 255) / 256)), dim3(256), 0, h->stream, keys2, vals2, rsl,
                       h->perm + (size_t)l * NS * 64, T, NS, nb);
    CKS(hipGetLastError());
    if ((rc = scan_u32(h, rsl, bsum, NS + 1)) || (!by_level && (rc = scan_u32(h, rpl, bsum, n)))) { cleanup(); return rc; }
    off1 += nl[l];
  }
  CKS(hipStreamSynchronize(h->stream));
  std::vector<unsigned> slots(L);
  for (int l = 0; l < L; ++l) CKS(hipMemcpy(&slots[l], h->rs + (size_t)l * (NS + 1) + NS, 4, hipMemcpyDeviceToHost));
  std::vector<unsigned long long> eb(L);
  h->n_slots = 0;
  for (int l = 0; l < L; ++l) {
    eb[l] = h->n_slots; h->n_slots += slots[l];
    if ((double)slots[l] < (double)nl[l]) { cleanup(); return fail(nullptr, VMR_EINVAL, "more than 2^32 report slots in one layer"); }
  }
  CKS(hipMalloc(&h->ebase, (size_t)L * 8));
  CKS(hipMemcpy(h->ebase, eb.data(), (size_t)L * 8, hipMemcpyHostToDevice));
  CKS(hipMalloc(&h->E, ((size_t)h->n_slots + SL_SLACK) * 4));
  CKS(hipMemsetAsync(h->E + h->n_slots, 0, (size_t)SL_SLACK * 4, h->stream));
  if (wide) {
    CKS(hipMalloc(&h->EX, ((size_t)h->n_slots + SL_SLACK) * 4));
    CKS(hipMemsetAsync(h->EX + h->n_slots, 0, (size_t)SL_SLACK * 4, h->stream));
  }
  unsigned long long off = 0;
  for (int l = 0; l < L; ++l) {
    unsigned* src = etmp_all ? etmp_all + off : etmp;
    unsigned* src2 = wide ? (etmp_all ? etmp2_all + off : etmp2) : nullptr;
    if (!etmp_all) (*fill)(l, rp + (size_t)l * n, etmp, etmp2);
    hipLaunchKernelGGL(k_sl_place, dim3(sgrid), dim3(256), 0, h->stream, h->perm + (size_t)l * NS * 64, rp + (size_t)l * n,
                       h->rs + (size_t)l * (NS + 1), src, h->E + eb[l], src2, wide ? h->EX + eb[l] : nullptr, h->sy + (size_t)l * NS, T, NS, g.Mp);
    off += nl[l];
  }
  CKS(hipGetLastError());
  CKS(hipStreamSynchronize(h->stream));
#undef CKS
  cleanup();
  return VMR_OK;
}

// ------------------------------------------------------------------------------------------
// the per-K sweep launchers (sweep_sl.hip, one object per K; a development build links only some of them)
// ------------------------------------------------------------------------------------------
#define SL_DECL(K_) __attribute__((weak)) int vmr_sl_launch_k##K_(vmr_ctx* h, int mode, const SlShape& sh, SlArgs& a);
SL_DECL(2) SL_DECL(3) SL_DECL(4) SL_DECL(5) SL_DECL(6) SL_DECL(7) SL_DECL(8)
#define SL_DECLB(K_) __attribute__((weak)) int vmr_sl_launch_batch_k##K_(vmr_ctx* h, hipStream_t st, int mode, int allfull, const SlUnit* units, const int* blk_unit, int nblocks, int tpb, size_t smem);
SL_DECLB(2) SL_DECLB(3) SL_DECLB(4) SL_DECLB(5) SL_DECLB(6) SL_DECLB(7) SL_DECLB(8)
sl_launch_batch_fn vmr_sl_batch_launcher(int K) {
  switch (K) {
    case 2: return vmr_sl_launch_batch_k2; case 3: return vmr_sl_launch_batch_k3; case 4: return vmr_sl_launch_batch_k4; case 5: return vmr_sl_launch_batch_k5;
    case 6: return vmr_sl_launch_batch_k6; case 7: return vmr_sl_launch_batch_k7; case 8: return vmr_sl_launch_batch_k8;
    default: return nullptr;
  }
}
sl_launch_fn vmr_sl_launcher(int K) {
  switch (K) {
    case 2: return vmr_sl_launch_k2; case 3: return vmr_sl_launch_k3; case 4: return vmr_sl_launch_k4; case 5: return vmr_sl_launch_k5;
    case 6: return vmr_sl_launch_k6; case 7: return vmr_sl_launch_k7; case 8: return vmr_sl_launch_k8;
    default: return nullptr;
  }
}
