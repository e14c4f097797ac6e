// Probe of gfx950's ds_read_b64_tr_b16 (LDS transpose read): prints which LDS element every (lane, result slot) receives when lane t of a
// 16-lane group supplies the address of 4 contiguous 16-bit elements.   hipcc --offload-arch=gfx950 tr16.hip -o tr16 && ./tr16
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned short* in, unsigned short* out, int rowstride) {
    __shared__ __attribute__((aligned(16))) unsigned short s[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) s[i] = in[i];
    __syncthreads();
    const int l = threadIdx.x, t = l & 15, g = l >> 4;
    const unsigned short* p = s + g * 1024 + (t >> 2) * rowstride + (t & 3) * 4;      // lane t: key t/4, channels 4(t%4)..+3
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
    unsigned short h[4096], o[256], *di, *dout;
    for (int i = 0; i < 4096; ++i) h[i] = (unsigned short)i;
    hipMalloc(&di, sizeof h), hipMalloc(&dout, sizeof o);
    hipMemcpy(di, h, sizeof h, hipMemcpyHostToDevice);
    for (int rs : {16, 32, 40}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout, rs);
        hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
        printf("rowstride %d (element index = group*1024 + key*rowstride + channel)\n", rs);
        int ok = 1;
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j) {
                const int want = (l >> 4) * 1024 + j * rs + (l & 15);      // hypothesis: lane gets channel (l&15) of keys 0..3
                if (o[l * 4 + j] != want) ok = 0;
            }
        printf("  hypothesis 'lane l slot j = key j, channel l&15': %s\n", ok ? "CONFIRMED" : "NO");
        if (!ok)
            for (int l = 0; l < 20; ++l) printf("  lane %2d: %5d %5d %5d %5d\n", l, o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3]);
    }
    return 0;
}
