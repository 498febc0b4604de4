// LDS tile images and transposing fragment reads shared by the MFMA attention and the
// weight-gradient kernels (gfx950).
//
// A [rows][DH] 16-bit (bf16 / fp16) tile has row stride DH*2 bytes and its 16-byte chunk index is XORed with
// swz<DH>(row), chosen so that BOTH fragment access patterns are bank-conflict free:
//   ds_read_b128      : 16 different rows, same chunk          -> 16 distinct 16-B slots
//   ds_read_b64_tr_b16: 4 rows x 64 B (two 16-lane groups)     -> 16 distinct 16-B slots
#pragma once
#include "common.h"

template <int DH> __device__ __forceinline__ int swz(int row);
template <> __device__ __forceinline__ int swz<128>(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
template <> __device__ __forceinline__ int swz<64>(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }
template <> __device__ __forceinline__ int swz<32>(int row) { return (row >> 2) & 3; }

// byte offset of (row, 16-B chunk c) in a tile
template <int DH> __device__ __forceinline__ int tile_off(int row, int c) {
  return row * (DH * 2) + ((c ^ swz<DH>(row)) << 4);
}

// ds_read_b64_tr_b16: the transposing read moves 16-bit patterns, whatever they encode
typedef short s16x4_lds_t __attribute__((ext_vector_type(4)));
template <typename HT>
__device__ __forceinline__ vec4<HT> tr_read(const char* p) {
  return __builtin_bit_cast(vec4<HT>, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_lds_t*)(p)));
}

template <typename HT>
__device__ __forceinline__ vec8<HT> cat8(vec4<HT> a, vec4<HT> b) {
  vec8<HT> r;
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
  r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
  return r;
}

template <typename HT>
__device__ __forceinline__ vec8<HT> pack8(const f32x16_t& v, int a) {
  vec8<HT> r;
#pragma unroll
  for (int u = 0; u < 8; ++u) r[u] = (HT)v[8 * a + u];
  return r;
}

template <typename HT>
__device__ __forceinline__ float dot8(vec8<HT> a, vec8<HT> b) {
  float s = 0.f;
#pragma unroll
  for (int u = 0; u < 8; ++u) s = fmaf((float)a[u], (float)b[u], s);
  return s;
}

// stage `nrows` rows (row r -> src + r*ld, DH bf16 each; rows >= nvalid are zero) into a tile
template <int DH, typename HT>
__device__ __forceinline__ void stage_tile(char* tile, const HT* src, size_t ld, int nrows, int nvalid, int tid,
                                           int nthreads) {
  constexpr int NC = DH / 8, UN = 8;
  // all of a thread's global loads are issued before the first LDS write (one HBM latency per tile, not one per chunk)
  for (int i0 = tid; i0 < nrows * NC; i0 += nthreads * UN) {
    vec8<HT> v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int idx = i0 + u * nthreads;
      const int row = idx / NC, c = idx % NC;
      if (idx < nrows * NC && row < nvalid) {
        v[u] = *reinterpret_cast<const vec8<HT>*>(src + (size_t)row * ld + c * 8);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[u][e] = (HT)0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int idx = i0 + u * nthreads;
      if (idx < nrows * NC) *reinterpret_cast<vec8<HT>*>(tile + tile_off<DH>(idx / NC, idx % NC)) = v[u];
    }
  }
}

// Two tiles (K and V of one head) staged together: every global load of BOTH tiles is issued before the first LDS write, so
// the block pays one memory latency for the pair instead of one per tile (the attention kernels are latency-bound at two
// workgroups per CU: every load issued earlier is time off the block's critical path).
template <int DH, typename HT>
__device__ __forceinline__ void stage_tile_pair(char* tile0, const HT* src0, char* tile1, const HT* src1, size_t ld, int nrows,
                                                int nvalid, int tid, int nthreads) {
  constexpr int NC = DH / 8, UN = 8;
  for (int i0 = tid; i0 < nrows * NC; i0 += nthreads * UN) {
    vec8<HT> v0[UN], v1[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int idx = i0 + u * nthreads;
      const int row = idx / NC, c = idx % NC;
      if (idx < nrows * NC && row < nvalid) {
        v0[u] = *reinterpret_cast<const vec8<HT>*>(src0 + (size_t)row * ld + c * 8);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v0[u][e] = (HT)0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int idx = i0 + u * nthreads;
      const int row = idx / NC, c = idx % NC;
      if (idx < nrows * NC && row < nvalid) {
        v1[u] = *reinterpret_cast<const vec8<HT>*>(src1 + (size_t)row * ld + c * 8);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v1[u][e] = (HT)0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int idx = i0 + u * nthreads;
      if (idx < nrows * NC) *reinterpret_cast<vec8<HT>*>(tile0 + tile_off<DH>(idx / NC, idx % NC)) = v0[u];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int idx = i0 + u * nthreads;
      if (idx < nrows * NC) *reinterpret_cast<vec8<HT>*>(tile1 + tile_off<DH>(idx / NC, idx % NC)) = v1[u];
    }
  }
}

// Block barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope fence as well: it waits for the
// acknowledgement of every global store the wave has in flight (microseconds behind a row of 16-bit gradient stores), which no
// phase boundary that hands over LDS contents needs.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// V^T / K^T fragment for the PV-style MFMA: lane (i = lane & 31 -> dh 32*db + i, g = lane >> 5)
// gets the 8 values tile[key(u)][dh], key(u) = kb + 8*(u>>2) + 4*g + (u&3)   (kb = 32*jb + 16*a)
template <int DH, typename HT>
__device__ __forceinline__ vec8<HT> tr_frag(const char* tile, int kb, int db, int lane) {
  const int gid = lane >> 4, p = lane & 15, g = gid >> 1;
  const int row0 = kb + 4 * g + (p >> 2);
  const int c = 4 * db + 2 * (gid & 1) + ((p & 3) >> 1);
  const int sub = (p & 1) * 8;
  vec4<HT> lo = tr_read<HT>(tile + tile_off<DH>(row0, c) + sub);
  vec4<HT> hi = tr_read<HT>(tile + tile_off<DH>(row0 + 8, c) + sub);
  return cat8<HT>(lo, hi);
}


// Row-per-lane stores.  In the 32x32 accumulator layout lane (row, g = lane >> 5) holds quad q of a 32-column block as
// columns 8q + 4g .. +3, so a row's 16 columns 16p .. 16p+15 (quads 2p, 2p+1) are interleaved between lanes l and l ^ 32.
// pair_exchange() trades one quad between the two lanes: afterwards g = 0 owns columns 16p .. 16p+7 and g = 1 owns
// 16p+8 .. 16p+15, i.e. ONE 16-byte bf16 store per lane instead of two 8-byte ones (the store phase of these kernels is
// bound by the number of store instructions, not by bytes).
// (gfx950: v_permlane32_swap trades the upper half of one register with the lower half of another - one VALU instruction per
//  quad element where a select + ds_bpermute + two selects stood)
__device__ __forceinline__ void pair_exchange(float (&out)[8], float e0, float e1, float e2, float e3, float o0, float o1,
                                              float o2, float o3, int g) {
  (void)g;
  const float e[4] = {e0, e1, e2, e3}, o[4] = {o0, o1, o2, o3};
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    // lanes 32 .. 63 of the first operand <-> lanes 0 .. 31 of the second:
    //   g = 0 lane: (e own, e of the partner)      g = 1 lane: (o of the partner, o own)
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(e[u]), __float_as_uint(o[u]), false, false);
    out[u] = __uint_as_float(r[0]);
    out[4 + u] = __uint_as_float(r[1]);
  }
}
// sum over the 16 lanes of a DPP row, in every lane of the row (rotations by 8, 4, 2, 1: four v_add_f32_dpp)
__device__ __forceinline__ float row16_sum(float x) {
  x += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), 0x128, 0xF, 0xF, false));
  x += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), 0x124, 0xF, 0xF, false));
  x += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), 0x122, 0xF, 0xF, false));
  x += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), 0x121, 0xF, 0xF, false));
  return x;
}
template <typename HT>
__device__ __forceinline__ void store8_h(HT* p, const float (&v)[8]) {
  vec8<HT> o;
  o[0] = (HT)v[0]; o[1] = (HT)v[1]; o[2] = (HT)v[2]; o[3] = (HT)v[3];
  o[4] = (HT)v[4]; o[5] = (HT)v[5]; o[6] = (HT)v[6]; o[7] = (HT)v[7];
  *reinterpret_cast<vec8<HT>*>(p) = o;
}

