// SOLA stitching of vocoder fragments (H17, reference TTS_infer_pack/TTS.py:1611-1637) for gfx950.
//
// For each neighbouring pair (f1, f2): cross-correlate the last `ov` samples of f1 with the first `ov` samples of f2
// (F.conv1d with padding ov/2, last output dropped), idx = first argmax; f1 loses its last ov-idx samples, f2 its
// first idx samples, and the first ov-idx samples of what is left of f2 are cross-faded with f1's tail under a periodic
// Hann window of length 2(ov-idx).  Everything (correlation, argmax, cross-fade, compaction) stays on the device;
// the argmax offsets never visit the host, only the final length does.
#include <vector>

#include "common.h"

namespace gsv {

// corr[j] = sum_k w1[j + k - ov/2] * w2[k], zero padded, j in [0, ov)   (one workgroup per output)
__global__ __launch_bounds__(256) void sola_corr_kernel(const float* __restrict__ w1, const float* __restrict__ w2, int ov,
                                                        float* __restrict__ corr) {
  __shared__ float red[4];
  const int j = blockIdx.x;
  const int off = j - ov / 2;
  float s = 0.f;
  for (int k = threadIdx.x; k < ov; k += 256) {
    const int i = off + k;
    if (i >= 0 && i < ov) s += w1[i] * w2[k];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) corr[j] = red[0] + red[1] + red[2] + red[3];
}

// first argmax of corr[0..ov) -> idx[pair]
__global__ __launch_bounds__(256) void sola_argmax_kernel(const float* __restrict__ corr, int ov, int* __restrict__ idx_out) {
  __shared__ float bv[256];
  __shared__ int bi[256];
  float best = -INFINITY;
  int b = 0x7fffffff;
  for (int j = threadIdx.x; j < ov; j += 256) {
    const float v = corr[j];
    if (v > best || (v == best && j < b)) { best = v; b = j; }
  }
  bv[threadIdx.x] = best; bi[threadIdx.x] = b;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      const float v = bv[threadIdx.x + o]; const int i2 = bi[threadIdx.x + o];
      if (v > bv[threadIdx.x] || (v == bv[threadIdx.x] && i2 < bi[threadIdx.x])) { bv[threadIdx.x] = v; bi[threadIdx.x] = i2; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *idx_out = bi[0] == 0x7fffffff ? 0 : bi[0];
}

// f2[idx + m] = win[m] * f2[idx + m] + win[n + m] * f1_tail[ov - n + m],  n = ov - idx, win = periodic Hann(2n)
__global__ void sola_blend_kernel(const float* __restrict__ f1_tail, float* __restrict__ f2, int ov, const int* __restrict__ idx_p) {
  const int idx = *idx_p, n = ov - idx;
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n) return;
  const float N = (float)(2 * n);
  const float wa = 0.5f - 0.5f * cosf(6.283185307179586f * (float)m / N);
  const float wb = 0.5f - 0.5f * cosf(6.283185307179586f * (float)(n + m) / N);
  f2[idx + m] = wa * f2[idx + m] + wb * f1_tail[ov - n + m];
}

// piece i = frag_i[start_i : len_i - trim_i], start_0 = 0, start_i = idx[i-1], trim_i = ov - idx[i] (0 for the last)
__global__ void sola_offsets_kernel(const int* __restrict__ lens, const int* __restrict__ idx, int n, int ov, long long* __restrict__ dst_off,
                                    int* __restrict__ total) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  long long o = 0;
  for (int i = 0; i < n; ++i) {
    dst_off[i] = o;
    const int start = i > 0 ? idx[i - 1] : 0;
    const int trim = i < n - 1 ? ov - idx[i] : 0;
    o += lens[i] - start - trim;
  }
  dst_off[n] = o;
  *total = (int)o;
}

__global__ void sola_gather_kernel(const float* __restrict__ frags, const long long* __restrict__ src_off, const int* __restrict__ idx,
                                   const long long* __restrict__ dst_off, int n, float* __restrict__ out) {
  const int i = blockIdx.y;
  const int start = i > 0 ? idx[i - 1] : 0;
  const long long cnt = dst_off[i + 1] - dst_off[i];
  const float* src = frags + src_off[i] + start;
  float* dst = out + dst_off[i];
  for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < cnt; k += (long long)gridDim.x * blockDim.x) dst[k] = src[k];
}

// ---- H13: peak normalisation + silence gaps + int16 conversion (reference TTS.py:1377-1429) --------------------------
// One workgroup per fragment, fragments already in output order.  Pass 1: peak = max|x| over the fragment (a NaN anywhere
// makes the reference's `max > 1` test false, so it disables the division here too).  Pass 2 (served from L2): the
// reference's arithmetic in the fragment's own dtype T -- x / peak in T when peak > 1, then * 32768 in T (numpy keeps
// float16 for float16 * int) -- truncated to int32 and wrapped to int16, followed by `gap` zero samples.
struct PostTable {
  const void* src[32];
  long long dst[32];
  int len[32];
};

template <typename T>
__global__ __launch_bounds__(1024) void postprocess_kernel(PostTable tab, int gap, short* __restrict__ out) {
  __shared__ float red[16];
  __shared__ int red_nan[16];
  const T* __restrict__ x = (const T*)tab.src[blockIdx.x];
  const int n = tab.len[blockIdx.x];
  short* __restrict__ o = out + tab.dst[blockIdx.x];
  float peak = 0.f;
  int nan = 0;
  for (int i = threadIdx.x; i < n; i += 1024) {
    const float v = fabsf((float)x[i]);
    nan |= (v != v);
    peak = fmaxf(peak, v);
  }
  peak = wave_max(peak);
  nan = __any(nan);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = peak; red_nan[threadIdx.x >> 6] = nan; }
  __syncthreads();
  peak = red[0];
  nan = red_nan[0];
  for (int w = 1; w < 16; ++w) { peak = fmaxf(peak, red[w]); nan |= red_nan[w]; }
  const bool divide = !nan && peak > 1.f;
  const T denom = (T)peak;                                     // exact: the peak is one of the fragment's values
  for (int i = threadIdx.x; i < n; i += 1024) {
    T v = x[i];
    if (divide) v = (T)((float)v / (float)denom);
    const T m = (T)((float)v * 32768.f);
    o[i] = (short)(int)(float)m;
  }
  for (int i = threadIdx.x; i < gap; i += 1024) o[n + i] = 0;
}

}  // namespace gsv

extern "C" int gsv_sola(float* frags, const int* lens, int n, int overlap, float* out, int* out_len, gsv_stream_t stream) {
  using namespace gsv;
  GSV_REQUIRE(frags && lens && out && out_len && n >= 1, "sola: bad argument");
  GSV_REQUIRE(overlap >= 2 && overlap % 2 == 0, "sola: overlap length %d must be even and >= 2", overlap);
  hipStream_t s = (hipStream_t)stream;
  std::vector<long long> off(n + 1, 0);
  for (int i = 0; i < n; ++i) {
    GSV_REQUIRE(lens[i] >= 2 * overlap || n == 1, "sola: fragment %d has %d samples, need >= 2 * overlap (%d)", i, lens[i], 2 * overlap);
    off[i + 1] = off[i] + lens[i];
  }
  // device scratch: corr[ov] | idx[n] | lens[n] | src_off[n+1] | dst_off[n+1] | total
  char* scratch = nullptr;
  const size_t b_corr = (size_t)overlap * 4, b_int = (size_t)n * 4, b_ll = (size_t)(n + 1) * 8;
  const size_t a_idx = (b_corr + 15) / 16 * 16, a_len = a_idx + (b_int + 15) / 16 * 16, a_src = a_len + (b_int + 15) / 16 * 16;
  const size_t a_dst = a_src + (b_ll + 15) / 16 * 16, a_tot = a_dst + (b_ll + 15) / 16 * 16, total_bytes = a_tot + 16;
  GSV_HIP(hipMalloc((void**)&scratch, total_bytes));
  float* corr = (float*)scratch;
  int* idx = (int*)(scratch + a_idx);
  int* dlens = (int*)(scratch + a_len);
  long long* src_off = (long long*)(scratch + a_src);
  long long* dst_off = (long long*)(scratch + a_dst);
  int* total = (int*)(scratch + a_tot);
  int rc = GSV_OK;
  auto fail = [&](hipError_t e, const char* what) { set_error("sola: %s -> %s", what, hipGetErrorString(e)); rc = GSV_ERR_HIP; };
  hipError_t e;
  if ((e = hipMemsetAsync(idx, 0, b_int, s)) != hipSuccess) fail(e, "memset");
  if (rc == GSV_OK && (e = hipMemcpyAsync(dlens, lens, b_int, hipMemcpyHostToDevice, s)) != hipSuccess) fail(e, "copy lens");
  if (rc == GSV_OK && (e = hipMemcpyAsync(src_off, off.data(), b_ll, hipMemcpyHostToDevice, s)) != hipSuccess) fail(e, "copy offsets");
  if (rc == GSV_OK && (e = hipStreamSynchronize(s)) != hipSuccess) fail(e, "sync");   // `off` is a stack vector
  for (int i = 0; rc == GSV_OK && i + 1 < n; ++i) {
    float* f1 = frags + off[i];
    float* f2 = frags + off[i + 1];
    const float* tail = f1 + lens[i] - overlap;
    hipLaunchKernelGGL(sola_corr_kernel, dim3(overlap), dim3(256), 0, s, tail, (const float*)f2, overlap, corr);
    hipLaunchKernelGGL(sola_argmax_kernel, dim3(1), dim3(256), 0, s, (const float*)corr, overlap, idx + i);
    hipLaunchKernelGGL(sola_blend_kernel, dim3(cdiv(overlap, 256)), dim3(256), 0, s, tail, f2, overlap, (const int*)(idx + i));
    if ((e = hipGetLastError()) != hipSuccess) fail(e, "launch");
  }
  if (rc == GSV_OK) {
    hipLaunchKernelGGL(sola_offsets_kernel, dim3(1), dim3(64), 0, s, (const int*)dlens, (const int*)idx, n, overlap, dst_off, total);
    hipLaunchKernelGGL(sola_gather_kernel, dim3(256, n), dim3(256), 0, s, (const float*)frags, (const long long*)src_off, (const int*)idx,
                       (const long long*)dst_off, n, out);
    if ((e = hipGetLastError()) != hipSuccess) fail(e, "launch");
  }
  if (rc == GSV_OK && (e = hipMemcpyAsync(out_len, total, 4, hipMemcpyDeviceToHost, s)) != hipSuccess) fail(e, "copy length");
  e = hipStreamSynchronize(s);
  if (rc == GSV_OK && e != hipSuccess) fail(e, "sync");
  (void)hipFree(scratch);
  return rc;
}


extern "C" int gsv_postprocess(const void* const* frags, const int* lens, int n, int dtype, int gap, int16_t* out,
                               gsv_stream_t stream) {
  using namespace gsv;
  GSV_REQUIRE(n >= 0 && gap >= 0 && (dtype == GSV_F16 || dtype == GSV_F32), "postprocess: bad argument");
  if (n == 0) return GSV_OK;
  GSV_REQUIRE(frags && lens && out, "postprocess: null pointer");
  hipStream_t s = (hipStream_t)stream;
  long long off = 0;
  for (int base = 0; base < n; base += 32) {
    PostTable tab{};
    const int m = n - base < 32 ? n - base : 32;
    for (int i = 0; i < m; ++i) {
      GSV_REQUIRE(lens[base + i] >= 0 && (lens[base + i] == 0 || frags[base + i]), "postprocess: fragment %d is null", base + i);
      tab.src[i] = frags[base + i];
      tab.len[i] = lens[base + i];
      tab.dst[i] = off;
      off += (long long)lens[base + i] + gap;
    }
    if (dtype == GSV_F16) hipLaunchKernelGGL(postprocess_kernel<_Float16>, dim3(m), dim3(1024), 0, s, tab, gap, (short*)out);
    else hipLaunchKernelGGL(postprocess_kernel<float>, dim3(m), dim3(1024), 0, s, tab, gap, (short*)out);
    GSV_HIP(hipGetLastError());
  }
  return GSV_OK;
}
