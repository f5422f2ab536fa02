// arranger_kernels.hip — GPU-side ragged -> padded arranger on the dense observation tensor.
//
// Replaces the reference's InOutArranger.rearrange_inputs / rearrange_outputs (DynEnv/models/models.py:219-250, :252-274:
// triple-nested Python over object arrays) with index kernels.  An observation row (env e, time t, agent a) holds, per
// object type i of a group, `count_i` valid rows of `feat_i` floats at a fixed offset (capacity `cap_i`).  With
// p = e*A + a (the reference chains environments player-major, models.py:222-223):
//   inputs[i]  [N_i][feat_i]  all objects of type i in (t, p, k) order               (rearrange_inputs :238-245)
//   slot[i]    [N_i]          row of that object in the padded tensor [T][maxCount][P][F] viewed as [T*maxCount*P][F]:
//                             (t*maxCount + sum_{i'<i} count_i'(t,p) + k) * P + p     (rearrange_outputs :259-268)
//   mask       [T][P][maxCount]  1 where slot >= objCounts[t][p]                      (ObsMask.createMask :166-180)
// HBM-bound by construction (every byte is touched once); no LDS tiles needed beyond the block scan.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dynenv.h"

#define ARR_BLOCK 256

struct ArrTypes {
  dynenv_arr_type_t t[DYNENV_ARR_MAX_TYPES];
  int n;
};

__device__ __forceinline__ int arr_count(const ArrTypes& ty, int i, const float* __restrict__ row, const int32_t* __restrict__ count_env, int e) {
  const dynenv_arr_type_t& d = ty.t[i];
  int c = d.count_value;
  if (d.count_mode == DYNENV_ARR_COUNT_ENV) c = count_env[(size_t)e * d.count_stride + d.count_index];
  else if (d.count_mode == DYNENV_ARR_COUNT_ROW) c = (int)row[d.count_index];
  c = c < 0 ? 0 : c;
  return c > d.cap ? d.cap : c;
}

// inclusive scan of one int per thread over a 256-thread block (4 waves): wave shuffles + one LDS hop
__device__ __forceinline__ int arr_block_scan(int v, int* waveTot /* [4] shared */, int& blockTotal) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  if (lane == 63) waveTot[w] = x;
  __syncthreads();
  int add = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < ARR_BLOCK / 64; ++k) { const int t = waveTot[k]; if (k < w) add += t; tot += t; }
  __syncthreads();
  blockTotal = tot;
  return x + add;
}

// K1: per (t, p): counts per type, objCounts, block-local exclusive prefix per type, block totals and block max
extern "C" __global__ void __launch_bounds__(ARR_BLOCK)
arr_count_kernel(const float* __restrict__ obs, int E, int T, int A, int D, ArrTypes ty, const int32_t* __restrict__ count_env,
                 int32_t* __restrict__ counts, int32_t* __restrict__ obj_counts, int32_t* __restrict__ base,
                 int32_t* __restrict__ blockTot /* [n+1][nBlocks]: per type totals, then block max of objCounts */) {
  __shared__ int waveTot[ARR_BLOCK / 64];
  const int P = E * A, TP = T * P, nBlocks = gridDim.x;
  const int tp = blockIdx.x * ARR_BLOCK + threadIdx.x;
  const bool live = tp < TP;
  const int t = live ? tp / P : 0, p = live ? tp % P : 0, e = p / A, a = p % A;
  const float* row = obs + (((size_t)e * T + t) * A + a) * D;
  int sum = 0;
  for (int i = 0; i < ty.n; ++i) {
    const int c = live ? arr_count(ty, i, row, count_env, e) : 0;
    sum += c;
    int tot;
    const int incl = arr_block_scan(c, waveTot, tot);
    if (live) { counts[(size_t)i * TP + tp] = c; base[(size_t)i * TP + tp] = incl - c; }
    if (threadIdx.x == 0) blockTot[(size_t)i * nBlocks + blockIdx.x] = tot;
  }
  if (live) obj_counts[tp] = sum;
  // block max of objCounts
  int m = sum;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { const int y = __shfl_xor(m, d, 64); m = y > m ? y : m; }
  if ((threadIdx.x & 63) == 0) waveTot[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    int mm = 0;
    for (int k = 0; k < ARR_BLOCK / 64; ++k) mm = waveTot[k] > mm ? waveTot[k] : mm;
    blockTot[(size_t)ty.n * nBlocks + blockIdx.x] = mm;
  }
}

// K2: one block: exclusive scan of the block totals per type, grand totals, global max
extern "C" __global__ void __launch_bounds__(ARR_BLOCK)
arr_scan_blocks_kernel(int32_t* __restrict__ blockTot, int nBlocks, int nTypes, int64_t* __restrict__ result /* [n] totals, [n] = max */) {
  __shared__ int waveTot[ARR_BLOCK / 64];
  for (int i = 0; i < nTypes; ++i) {
    int carry = 0;
    for (int b0 = 0; b0 < nBlocks; b0 += ARR_BLOCK) {
      const int b = b0 + threadIdx.x;
      const int v = b < nBlocks ? blockTot[(size_t)i * nBlocks + b] : 0;
      int tot;
      const int incl = arr_block_scan(v, waveTot, tot);
      if (b < nBlocks) blockTot[(size_t)i * nBlocks + b] = carry + incl - v;
      carry += tot;
    }
    if (threadIdx.x == 0) result[i] = carry;
  }
  int m = 0;
  for (int b = threadIdx.x; b < nBlocks; b += ARR_BLOCK) { const int v = blockTot[(size_t)nTypes * nBlocks + b]; m = v > m ? v : m; }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { const int y = __shfl_xor(m, d, 64); m = y > m ? y : m; }
  if ((threadIdx.x & 63) == 0) waveTot[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    int mm = 0;
    for (int k = 0; k < ARR_BLOCK / 64; ++k) mm = waveTot[k] > mm ? waveTot[k] : mm;
    result[nTypes] = mm;
  }
}

// K2b: base += offset of its block
extern "C" __global__ void __launch_bounds__(ARR_BLOCK)
arr_add_offsets_kernel(int32_t* __restrict__ base, const int32_t* __restrict__ blockTot, int TP, int nBlocks, int nTypes) {
  const int tp = blockIdx.x * ARR_BLOCK + threadIdx.x;
  if (tp >= TP) return;
  for (int i = 0; i < nTypes; ++i) base[(size_t)i * TP + tp] += blockTot[(size_t)i * nBlocks + blockIdx.x];
}

// K3: one thread per (t, p, j): j-th object slot of the player's padded list
extern "C" __global__ void __launch_bounds__(ARR_BLOCK)
arr_gather_kernel(const float* __restrict__ obs, int E, int T, int A, int D, ArrTypes ty, const int32_t* __restrict__ counts,
                  const int32_t* __restrict__ base, int maxCount, int jSpan /* = max(capSum, maxCount) */,
                  float* in0, float* in1, float* in2, float* in3, int32_t* sl0, int32_t* sl1, int32_t* sl2, int32_t* sl3,
                  uint8_t* __restrict__ mask) {
  const int P = E * A, TP = T * P;
  const long long gid = (long long)blockIdx.x * ARR_BLOCK + threadIdx.x;
  const int tp = (int)(gid / jSpan), j = (int)(gid % jSpan);
  if (tp >= TP) return;
  const int t = tp / P, p = tp % P, e = p / A, a = p % A;
  int off = 0, type = -1, k = 0;
  for (int i = 0; i < ty.n; ++i) {
    const int c = counts[(size_t)i * TP + tp];
    if (type < 0 && j < off + c) { type = i; k = j - off; }
    off += c;
  }
  if (mask && j < maxCount) mask[(size_t)tp * maxCount + j] = (uint8_t)(j >= off);
  if (type < 0) return;
  const dynenv_arr_type_t d = ty.t[type];
  const float* src = obs + (((size_t)e * T + t) * A + a) * D + d.offset + (size_t)k * d.feat;
  const size_t n = (size_t)base[(size_t)type * TP + tp] + k;
  float* in = type == 0 ? in0 : type == 1 ? in1 : type == 2 ? in2 : in3;
  int32_t* sl = type == 0 ? sl0 : type == 1 ? sl1 : type == 2 ? sl2 : sl3;
  if (in) { float* dst = in + n * d.feat; for (int f = 0; f < d.feat; ++f) dst[f] = src[f]; }
  if (sl) sl[n] = (int32_t)(((long long)t * maxCount + j) * P + p);
}

// K4: padded[slot[n]][:] = emb[n][:]   (one thread per float)
extern "C" __global__ void __launch_bounds__(ARR_BLOCK)
arr_scatter_kernel(const float* __restrict__ emb, const int32_t* __restrict__ slot, long long N, int F, float* __restrict__ padded) {
  const long long gid = (long long)blockIdx.x * ARR_BLOCK + threadIdx.x;
  if (gid >= N * F) return;
  const long long n = gid / F;
  const int f = (int)(gid % F);
  padded[(size_t)slot[n] * F + f] = emb[gid];
}

// K5: the padded tensor in one pass, no pre-zeroing: one thread per float4 COLUMN (t, p, f) of padded[t][:][p][:] walking down its
// player's slots j = 0 .. maxCount-1 (round 1-4: one thread per slot, 3.8-4.4 TB/s; this form 5.0-5.7 TB/s): the counts and bases
// of the player are read once (not once per slot), the index arithmetic is one add per slot, and four independent 16-byte loads
// are in flight per thread before their stores.  A wave still writes 1 KB contiguous per slot; the reads walk the player's
// consecutive embedding rows.  (HBM-bound: the padded tensor is written once, every embedding read once.)
// (the padded tensor is written once and not read here: non-temporal stores - worth 15-25 % where most slots are padding zeros, nothing
//  where most carry an embedding; non-temporal LOADS of the embeddings gain 5-10 % on dense inputs and lose 17 % on sparse ones: not used)
typedef float arr_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void arr_st_nt(float4* p, float4 v) { arr_f4 w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, reinterpret_cast<arr_f4*>(p)); }
#define ARR_ST(p, v) arr_st_nt((p), (v))
extern "C" __global__ void __launch_bounds__(ARR_BLOCK)
arr_pad_cols_kernel(const float* __restrict__ e0, const float* __restrict__ e1, const float* __restrict__ e2, const float* __restrict__ e3,
                    const int32_t* __restrict__ counts, const int32_t* __restrict__ base, int nTypes, int T, int P, int maxCount, int F4,
                    float4* __restrict__ padded) {
  const unsigned x = blockIdx.x * ARR_BLOCK + threadIdx.x;
  if (x >= (unsigned)P * (unsigned)F4) return;
  const int p = (int)(x / (unsigned)F4), f = (int)(x % (unsigned)F4);
  const int t = (int)blockIdx.y, TP = T * P, tp = t * P + p;
  const size_t stride = (size_t)P * F4;  // float4s between consecutive slots of one column
  float4* out = padded + (size_t)t * maxCount * stride + x;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  int j = 0;
  for (int i = 0; i < nTypes && j < maxCount; ++i) {
    int c = counts[(size_t)i * TP + tp];
    if (c > maxCount - j) c = maxCount - j;  // (cannot happen: maxCount is the largest total; keeps the writes in bounds whatever the counts hold)
    const float* e = i == 0 ? e0 : i == 1 ? e1 : i == 2 ? e2 : e3;
    if (e) {
      const float4* src = reinterpret_cast<const float4*>(e) + (size_t)base[(size_t)i * TP + tp] * F4 + f;
      int k = 0;
      for (; k + 4 <= c; k += 4) {
        const float4 v0 = src[(size_t)(k + 0) * F4], v1 = src[(size_t)(k + 1) * F4], v2 = src[(size_t)(k + 2) * F4], v3 = src[(size_t)(k + 3) * F4];
        ARR_ST(out, v0); ARR_ST(out + stride, v1); ARR_ST(out + 2 * stride, v2); ARR_ST(out + 3 * stride, v3);
        out += 4 * stride;
      }
      for (; k < c; ++k) { ARR_ST(out, src[(size_t)k * F4]); out += stride; }
    } else {
      for (int k = 0; k < c; ++k) { ARR_ST(out, zero); out += stride; }
    }
    j += c;
  }
  for (; j < maxCount; ++j) { ARR_ST(out, zero); out += stride; }
}

// ------------------------------------------------------------------------------------------------
// De-duplicated transport format of an observation tensor whose agent rows share a tail (Driving Full: obstacles,
// pedestrians and lane rows are the same 160 floats for all 10 agents of an environment): per (env, time) the A agent
// prefixes of `split` floats, then the tail of D - split floats once.  Used around the multi-GPU all-gather (2.6x fewer
// bytes over xGMI); unpack restores the dense tensor bit for bit.
// ------------------------------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(ARR_BLOCK)
obs_pack_kernel(const float* __restrict__ obs, long long nET, int A, int D, int split, float* __restrict__ packed) {
  const int P = A * split + (D - split);
  const long long gid = (long long)blockIdx.x * ARR_BLOCK + threadIdx.x;
  if (gid >= nET * P) return;
  const long long et = gid / P;
  const int k = (int)(gid % P);
  const float* row = obs + (size_t)et * A * D;
  packed[gid] = k < A * split ? row[(size_t)(k / split) * D + (k % split)] : row[split + (k - A * split)];
}
extern "C" __global__ void __launch_bounds__(ARR_BLOCK)
obs_unpack_kernel(const float* __restrict__ packed, long long nET, int A, int D, int split, float* __restrict__ obs,
                  long long srcStride /* floats between the packed blocks of consecutive ranks (blockIdx.y) */) {
  const int P = A * split + (D - split);
  const long long gid = (long long)blockIdx.x * ARR_BLOCK + threadIdx.x;
  if (gid >= nET * A * D) return;
  const long long et = gid / ((long long)A * D);
  const int r = (int)(gid % ((long long)A * D)), a = r / D, f = r % D;
  const float* src = packed + (size_t)blockIdx.y * srcStride + (size_t)et * P;
  obs[(size_t)blockIdx.y * nET * A * D + gid] = f < split ? src[a * split + f] : src[A * split + (f - split)];
}

// ------------------------------------------------------------------------------------------------
// Peer-compacted transport format of a Driving Full observation (DrivingEnvironment.getFullState :686-747): agent a's row
// is [ self 9 | the other A-1 cars, 7 floats each, in car order without a | tail ], and the 7 floats a row holds of car c
// are columns {0..5, 8} of c's own self block (position, cos/sin, size, finished).  An (env, time) is therefore fully
// described by the A self blocks and the tail once: 9A + (D - 9 - 7(A-1)) floats instead of A*D (A = 10: 250 vs 2320,
// 9.3x fewer bytes through the all-gather).  Expansion copies floats, so the dense tensor comes back bit for bit.
// ------------------------------------------------------------------------------------------------
#define PEER_SELF 9
#define PEER_COLS 7
__device__ __forceinline__ int peer_src_index(int a, int f, int A, int carsEnd) {
  if (f < PEER_SELF) return a * PEER_SELF + f;
  if (f >= carsEnd) return A * PEER_SELF + (f - carsEnd);
  int c = (f - PEER_SELF) / PEER_COLS;
  const int kk = (f - PEER_SELF) - c * PEER_COLS;
  c += (c >= a);
  return c * PEER_SELF + (kk < 6 ? kk : 8);
}
extern "C" __global__ void __launch_bounds__(ARR_BLOCK)
obs_pack_peers_kernel(const float* __restrict__ obs, long long nET, int A, int D, float* __restrict__ packed) {
  const int carsEnd = PEER_SELF + (A - 1) * PEER_COLS;
  const int P = A * PEER_SELF + (D - carsEnd);
  const long long gid = (long long)blockIdx.x * ARR_BLOCK + threadIdx.x;
  if (gid >= nET * P) return;
  const long long et = gid / P;
  const int k = (int)(gid % P);
  const float* row = obs + (size_t)et * A * D;
  packed[gid] = k < A * PEER_SELF ? row[(size_t)(k / PEER_SELF) * D + (k % PEER_SELF)] : row[carsEnd + (k - A * PEER_SELF)];
}
// one thread per 4 consecutive floats of the dense tensor when D % 4 == 0 (vec = 4), else per float (vec = 1)
template <int VEC>
__device__ __forceinline__ void obs_unpack_peers_body(const float* __restrict__ packed, long long nET, int A, int D,
                                                      float* __restrict__ obs, long long srcStride) {
  const int carsEnd = PEER_SELF + (A - 1) * PEER_COLS;
  const int P = A * PEER_SELF + (D - carsEnd);
  const long long gid = ((long long)blockIdx.x * ARR_BLOCK + threadIdx.x) * VEC;
  const long long rowLen = (long long)A * D;
  if (gid >= nET * rowLen) return;
  const long long et = gid / rowLen;
  const int r = (int)(gid % rowLen), a = r / D, f = r % D;
  const float* src = packed + (size_t)blockIdx.y * srcStride + (size_t)et * P;
  float* dst = obs + (size_t)blockIdx.y * nET * rowLen + gid;
  if (VEC == 4) {
    float4 v;
    v.x = src[peer_src_index(a, f + 0, A, carsEnd)]; v.y = src[peer_src_index(a, f + 1, A, carsEnd)];
    v.z = src[peer_src_index(a, f + 2, A, carsEnd)]; v.w = src[peer_src_index(a, f + 3, A, carsEnd)];
    *reinterpret_cast<float4*>(dst) = v;
  } else {
    *dst = src[peer_src_index(a, f, A, carsEnd)];
  }
}
// The expansion as a row job: the map (agent, feature) -> index into the compacted row is the same for every (env, time), so a
// block builds it ONCE in LDS (16-bit entries), then for each of its rows stages the compacted floats (<= 304) in LDS and streams
// the dense row out as float4s, row after row (contiguous 9 KB bursts) - ~10 instructions per 16 bytes written instead of a 64-bit
// division, two 32-bit ones and four index computations per thread.  D % 4 == 0, A <= PEER_ROWS_MAXA; one block serves
// `rowsPerBlock` consecutive rows of one rank.
// (Measured, 8 ranks x 4096 environments, 304 MB written; a memset of that size takes 43 us: 95 us for the per-thread index
//  arithmetic, 75 us for this form; 81 us with 320 threads (two whole passes over a row's 580 float4 columns instead of 2.3);
//  99 us when a thread keeps its column's indices in registers and walks down the rows - 1 KB bursts 9 KB apart.)
#define PEER_ROWS_MAXA 16
#define PEER_ROWS_MAXD (PEER_SELF + (PEER_ROWS_MAXA - 1) * PEER_COLS + 160)
#define PEER_ROWS_MAXP (PEER_ROWS_MAXA * PEER_SELF + 160 + 8)
#ifndef PEER_ROWS_BATCH
#define PEER_ROWS_BATCH 4  /* rows expanded between two barriers (1 x 580 float4 columns over 256 threads leave 24 % of the last pass idle): 64.5 / 62.3 / 61.3 us for 1 / 2 / 4 */
#endif
extern "C" __global__ void __launch_bounds__(1024)
obs_unpack_peers_rows_kernel(const float* __restrict__ packed, long long nET, int A, int D, float* __restrict__ obs, long long srcStride,
                             int rowsPerBlock) {
  __shared__ __align__(8) unsigned short tbl[PEER_ROWS_MAXA * PEER_ROWS_MAXD];
  __shared__ float stage[2][PEER_ROWS_BATCH * PEER_ROWS_MAXP];
  const int carsEnd = PEER_SELF + (A - 1) * PEER_COLS;
  const int P = A * PEER_SELF + (D - carsEnd);
  const int rowLen = A * D, nv = rowLen >> 2;
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int i = tid; i < rowLen; i += nt) tbl[i] = (unsigned short)peer_src_index(i / D, i % D, A, carsEnd);
  const long long et0 = (long long)blockIdx.x * rowsPerBlock;
  const float* srcBase = packed + (size_t)blockIdx.y * srcStride;
  float* dstBase = obs + (size_t)blockIdx.y * nET * rowLen;
  int nRows = rowsPerBlock;
  if (et0 + nRows > nET) nRows = (int)(nET - et0);
  // (the compacted rows of a block are contiguous in the source: a batch of B rows is B * P consecutive floats)
  const int first = nRows < PEER_ROWS_BATCH ? nRows : PEER_ROWS_BATCH;
  for (int i = tid; i < first * P; i += nt) stage[0][(i / P) * PEER_ROWS_MAXP + (i % P)] = srcBase[(size_t)et0 * P + i];
  __syncthreads();
  for (int e = 0, bi = 0; e < nRows; e += PEER_ROWS_BATCH, ++bi) {
    const float* cur = stage[bi & 1];
    const int nb = nRows - e < PEER_ROWS_BATCH ? nRows - e : PEER_ROWS_BATCH;
    const int nn = nRows - (e + PEER_ROWS_BATCH) < PEER_ROWS_BATCH ? nRows - (e + PEER_ROWS_BATCH) : PEER_ROWS_BATCH;
    if (nn > 0)  // the next batch's compacted floats travel while this one is written out
      for (int i = tid; i < nn * P; i += nt) stage[(bi + 1) & 1][(i / P) * PEER_ROWS_MAXP + (i % P)] = srcBase[(size_t)(et0 + e + PEER_ROWS_BATCH) * P + i];
    float4* dst = reinterpret_cast<float4*>(dstBase + (size_t)(et0 + e) * rowLen);
    for (int v = tid; v < nb * nv; v += nt) {
      const int r = v >= nv ? v / nv : 0, c = v - r * nv;
      const ushort4 ix = reinterpret_cast<const ushort4*>(tbl)[c];
      const float* row = cur + r * PEER_ROWS_MAXP;
      float4 o;
      o.x = row[ix.x]; o.y = row[ix.y]; o.z = row[ix.z]; o.w = row[ix.w];
      arr_st_nt(dst + v, o);  // (written once, read by somebody else later: non-temporal, 77 -> 65 us for 304 MB)
    }
    __syncthreads();
  }
}
extern "C" __global__ void __launch_bounds__(ARR_BLOCK)
obs_unpack_peers4_kernel(const float* __restrict__ packed, long long nET, int A, int D, float* __restrict__ obs, long long srcStride) {
  obs_unpack_peers_body<4>(packed, nET, A, D, obs, srcStride);
}
extern "C" __global__ void __launch_bounds__(ARR_BLOCK)
obs_unpack_peers1_kernel(const float* __restrict__ packed, long long nET, int A, int D, float* __restrict__ obs, long long srcStride) {
  obs_unpack_peers_body<1>(packed, nET, A, D, obs, srcStride);
}
