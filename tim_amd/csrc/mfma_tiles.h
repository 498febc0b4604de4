// LDS tile images and transposing fragment reads shared by the MFMA attention and the
// weight-gradient kernels (gfx950).
//
// A [rows][DH] bf16 tile has row stride DH*2 bytes and its 16-byte chunk index is XORed with
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

typedef __attribute__((address_space(3))) bf16x4_t* lds_b64_ptr;

__device__ __forceinline__ bf16x4_t tr_read(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b64_ptr)(p));
}

__device__ __forceinline__ bf16x8_t cat8(bf16x4_t a, bf16x4_t b) {
  bf16x8_t r;
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
  r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
  return r;
}

__device__ __forceinline__ bf16x8_t pack8(const f32x16_t& v, int a) {
  bf16x8_t r;
#pragma unroll
  for (int u = 0; u < 8; ++u) r[u] = (bf16_t)v[8 * a + u];
  return r;
}

__device__ __forceinline__ float dot8(bf16x8_t a, bf16x8_t b) {
  float s = 0.f;
#pragma unroll
  for (int u = 0; u < 8; ++u) s = fmaf((float)a[u], (float)b[u], s);
  return s;
}

// stage `nrows` rows (row r -> src + r*ld, DH bf16 each; rows >= nvalid are zero) into a tile
template <int DH>
__device__ __forceinline__ void stage_tile(char* tile, const bf16_t* src, size_t ld, int nrows, int nvalid, int tid,
                                           int nthreads) {
  constexpr int NC = DH / 8;
  for (int idx = tid; idx < nrows * NC; idx += nthreads) {
    const int row = idx / NC, c = idx % NC;
    bf16x8_t v;
    if (row < nvalid) {
      v = *reinterpret_cast<const bf16x8_t*>(src + (size_t)row * ld + c * 8);
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = (bf16_t)0.f;
    }
    *reinterpret_cast<bf16x8_t*>(tile + tile_off<DH>(row, c)) = v;
  }
}

// V^T / K^T fragment for the PV-style MFMA: lane (i = lane & 31 -> dh 32*db + i, g = lane >> 5)
// gets the 8 values tile[key(u)][dh], key(u) = kb + 8*(u>>2) + 4*g + (u&3)   (kb = 32*jb + 16*a)
template <int DH>
__device__ __forceinline__ bf16x8_t tr_frag(const char* tile, int kb, int db, int lane) {
  const int gid = lane >> 4, p = lane & 15, g = gid >> 1;
  const int row0 = kb + 4 * g + (p >> 2);
  const int c = 4 * db + 2 * (gid & 1) + ((p & 3) >> 1);
  const int sub = (p & 1) * 8;
  bf16x4_t lo = tr_read(tile + tile_off<DH>(row0, c) + sub);
  bf16x4_t hi = tr_read(tile + tile_off<DH>(row0 + 8, c) + sub);
  return cat8(lo, hi);
}

