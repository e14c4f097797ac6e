// Shared pieces of the fp32-MFMA 3-D convolution kernels (conv3d_fwd.hip, conv3d.hip).
#pragma once
#include "common.h"

namespace mvsconv {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using rsrc_t = __amdgpu_buffer_rsrc_t;

constexpr int NWAVES = 4;
constexpr unsigned OOB = 0x80000000u;     // buffer offset beyond every descriptor range used here => load returns 0

// wave-uniform buffer descriptor: loads beyond `bytes` (or with OOB as offset) return 0, which is how zero padding,
// the tile halo outside the volume and channel padding are produced without branches
__device__ __forceinline__ rsrc_t make_rsrc(const float* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load(rsrc_t r, unsigned voff_bytes, unsigned soff_bytes) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff_bytes, soff_bytes, 0));
}

constexpr int np_of(int NT) { return NT == 1 ? 16 : (NT == 2 ? 48 : 80); }   // packed cout row, == 16 (mod 32)
constexpr int pad_cs(int raw, int shw) {
    // channel stride of the LDS input tile: == 16 (mod 32) for unit-stride fragment reads, odd for stride-2 reads
    return shw == 1 ? raw + ((16 - raw % 32) + 32) % 32 : raw + ((raw % 2 == 0) ? 1 : 0);
}
__host__ __device__ inline int nt_of(int Cout) { int nt = (Cout + 15) / 16; return nt == 3 ? 4 : nt; }

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// fused epilogue for one lane's 4 consecutive output voxels of one channel
__device__ __forceinline__ f32x4 bn_act(f32x4 a, float sc, float sh, int relu) {
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float v = fmaf(a[r], sc, sh);
        o[r] = relu ? fmaxf(v, 0.0f) : v;
    }
    return o;
}

// XCD-aware block order (cdna_hip_programming.md T1): the dispatcher sends consecutive workgroup ids round-robin to the 8
// XCDs, each with its own L2; remap so every XCD works on one contiguous slab of the (x, y, z) block grid and the halo
// rows shared by neighbouring tiles hit in the same L2.  Bijective for any grid size.  Speed only, never correctness.
__device__ __forceinline__ void xcd_block_coords(unsigned& bx, unsigned& by, unsigned& bz) {
    const unsigned gx = gridDim.x, gy = gridDim.y, nb = gx * gy * gridDim.z;
    const unsigned id = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned q = nb / 8, r = nb % 8, xcd = id % 8, loc = id / 8;
    const unsigned nid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    bx = nid % gx;
    by = (nid / gx) % gy;
    bz = nid / (gx * gy);
}

// The same remap as a linear id (for kernels that choose their own decomposition of it).
__device__ __forceinline__ unsigned xcd_linear_block_id() {
    const unsigned gx = gridDim.x, gy = gridDim.y, nb = gx * gy * gridDim.z;
    const unsigned id = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned q = nb / 8, r = nb % 8, xcd = id % 8, loc = id / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}
inline int check_conv_args(const char* who, int B, int Cin, int Cout, int Di, int Hi, int Wi) {
    MVS_REQUIRE(B >= 1 && Di >= 1 && Hi >= 1 && Wi >= 1, "%s: bad shape B=%d D=%d H=%d W=%d", who, B, Di, Hi, Wi);
    MVS_REQUIRE(Cin >= 4 && Cin % 4 == 0, "%s: Cin must be a multiple of 4 (got %d)", who, Cin);
    MVS_REQUIRE(Cout >= 8 && Cout % 8 == 0 && Cout <= 64, "%s: Cout must be a multiple of 8, <= 64 (got %d)", who, Cout);
    MVS_REQUIRE((int64_t)8 * Di * Hi * Wi * 4 < ((int64_t)1 << 31), "%s: 8 input channels exceed the 2 GiB buffer window", who);
    return MVS_OK;
}

// grouped stride-(1,2,2) transposed convolution (deconv3d_s1.hip)
bool deconv_s1_supported(int Cout);
int64_t deconv_s1_packed_floats(int Cin, int Cout);
int deconv_s1_pack(const float* w, int Cin, int Cout, float* out, hipStream_t s);
int deconv_s1_launch(const float* x, const float* wp, const float* scale, const float* shift, const float* res, float* y, int B,
                     int Cin, int Cout, int Di, int Hi, int Wi, int relu, hipStream_t s);

int deconv_s1_prob_launch(const float* x, const float* wp, const float* scale, const float* shift, const float* res,
                          const float* prob_w, const float* prob_b, float* logits, int B, int Cin, int Di, int Hi, int Wi, int relu,
                          hipStream_t s);

}  // namespace mvsconv
