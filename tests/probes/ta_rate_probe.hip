// How fast does a CU's vector-memory path deliver MFMA A-FRAGMENTS straight from L2 into registers?  ("A fragments
// loaded L2 -> registers with global_load_dwordx4, B stays in LDS".)
//
// An A fragment of v_mfma_f32_32x32x16_f16 is fixed by the hardware: lane (lr = lane & 31, lk = lane >> 5) holds 8 fp16 = 16 bytes of
// row lr, k-half lk.  Rows are output pixels; in the zero-bordered NHWC operand consecutive pixels are C * 2 = 512 bytes apart (256
// channels), so the 64 lanes of one global_load_dwordx4 touch 32 different 128-byte lines, 16 bytes each from two lanes that are 32
// lanes apart - NO two adjacent lanes are contiguous.  The LDS-DMA staging of conv_igemm_dw reads the same bytes with lane -> (row lane >> 2,
// 16-byte slot lane & 3): every quad of lanes covers 64 contiguous bytes.
//
// This probe times both address patterns on an L2-resident window, eight waves per CU (the occupancy of conv_igemm_dw), and prints the
// shader cycles one CU needs per load instruction - to be held against the k-tile budget of the convolution: 32 fragment loads per k-tile
// and CU (8 waves x 2 sets x 2 row tiles) inside ~1 300 cycles.
//   hipcc --offload-arch=gfx950 -O3 -o ta_rate_probe tests/probes/ta_rate_probe.hip && ./ta_rate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// PATTERN 0: fragment addressing (lane -> pixel lr, 16-byte slot s * 2 + lk of the pixel's 64-byte channel slice; 4 loads = 2 row tiles x 2 sets)
// PATTERN 1: quad-contiguous addressing of the same bytes (lane -> pixel lane >> 2, slot lane & 3; 4 loads = 4 x 16 pixels)
template <int PATTERN>
__global__ __launch_bounds__(512, 1) void probe(const char* __restrict__ x, int pixel_bytes, int iters, unsigned* out, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 31, lk = lane >> 5;
    // every workgroup walks its own 256-pixel window; a k-tile = one 64-byte channel slice of the window at one of 9 tap offsets
    const char* base = x + (size_t)blockIdx.x * 256 * pixel_bytes + (size_t)(pixel_bytes * 34);
    const int wr = wave >> 1;
    u32x4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const int tap = it % 9, slice = (it / 9) % (pixel_bytes / 64);
        const long long toff = (long long)((tap / 3 - 1) * 33 + (tap % 3 - 1)) * pixel_bytes + slice * 64;
        u32x4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            long long off;
            if (PATTERN == 0) off = (long long)(wr * 64 + (q >> 1) * 32 + lr) * pixel_bytes + ((q & 1) * 2 + lk) * 16;
            else off = (long long)(wr * 64 + q * 16 + (lane >> 2)) * pixel_bytes + (lane & 3) * 16;
            v[q] = *reinterpret_cast<const u32x4*>(base + toff + off);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc ^= v[q];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + threadIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    const int pixel_bytes = 512, blocks = 256, iters = 2304;    // 256 channels of fp16; 9 taps x 8 slices x 32 = the k-loop of 256 -> 256 3x3, x 32
    const size_t bytes = (size_t)(blocks * 256 + 128) * pixel_bytes;   // 33.6 MB: inside the 256 MB MALL, 4 MB per XCD's L2 window at a time
    char* x;
    unsigned* out;
    unsigned long long* cyc;
    hipMalloc(&x, bytes);
    hipMemset(x, 1, bytes);
    hipMalloc(&out, blocks * 512 * 4);
    hipMalloc(&cyc, blocks * 8);
    std::vector<unsigned long long> h(blocks);
    const char* names[2] = {"fragment addressing (lane -> pixel lr, slot s*2+lk)", "quad-contiguous addressing (lane -> pixel lane>>2, slot lane&3)"};
    for (int rep = 0; rep < 2; ++rep)
        for (int pat = 0; pat < 2; ++pat) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipEventRecord(e0);
            if (pat == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(512), 0, 0, x, pixel_bytes, iters, out, cyc);
            else hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(512), 0, 0, x, pixel_bytes, iters, out, cyc);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
            double s = 0;
            for (auto c : h) s += (double)c;
            // s_memtime ticks at 100 MHz on gfx94x / gfx950: convert through the wall time instead - cycles = ms x sclk is left to the reader;
            // the figure that matters is time per (CU, k-tile) = ms / iters against the convolution's ~0.73 us per k-tile
            const double us_per_ktile = ms * 1e3 / iters;
            const double gbs = (double)blocks * iters * 8 * 4 * 1024 / (ms * 1e-3) / 1e9;
            printf("%-66s %8.3f ms  %6.3f us per (CU, k-tile of 32 loads = 32 KB)  %7.0f GB/s chip  (memtime ticks per block: %.0f)\n", names[pat], ms,
                   us_per_ktile, gbs, s / blocks);
        }
    printf("conv_igemm_dw spends ~1 317 cycles = 0.73 us at 1.8 GHz per k-tile; fragment loads of A would need the first figure INSIDE that, next to 16 KB of weight LDS-DMA\n");
    return 0;
}
