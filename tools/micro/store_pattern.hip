// Micro-benchmark: does the epilogue's store shape matter?  A [M][C] bf16 matrix is written (and, in the +res variants, a
// second one read) by 256-row x C tiles, one per block of 512 threads, either (a) in the conv epilogue's shape -- lane
// (l31, hh) owns a row of the wave's 32 and stores 16 B per instruction, 32 B per row per instruction -- or (b) as whole rows:
// every wave instruction stores 1 KiB of consecutive bytes.  Prints GB/s of each.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, bool RES>
__global__ __launch_bounds__(512) void k(const unsigned short* __restrict__ res, unsigned short* __restrict__ out, int M, int C, int tiles) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int m0 = t * 256;
        if (MODE == 0) {
            // wave (wm = wave % 4, wn = wave / 4): rows m0 + wm*64 + j*32 + l31, cols wn*(C/2) + ob*32 + 16 p + 8 hh
            const int wm = wave & 3, wn = wave >> 2;
            for (int j = 0; j < 2; ++j) {
                const long long row = m0 + wm * 64 + j * 32 + l31;
                for (int ob = 0; ob < C / 64; ++ob)
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const int col = wn * (C / 2) + ob * 32 + 16 * p + 8 * hh;
                        u32x4 v = {(unsigned)row, (unsigned)col, 1u, 2u};
                        if (RES) { u32x4 r = *(const u32x4*)(res + row * C + col); v += r; }
                        *(u32x4*)(out + row * C + col) = v;
                    }
            }
        } else {
            // whole rows: chunk index c over the tile's 256*C/8 16-B chunks, consecutive lanes consecutive chunks
            const int nchunk = 256 * C / 8;
            for (int c = tid; c < nchunk; c += 512) {
                const long long off = (long long)m0 * C + (long long)c * 8;
                u32x4 v = {(unsigned)c, 3u, 1u, 2u};
                if (RES) { u32x4 r = *(const u32x4*)(res + off); v += r; }
                *(u32x4*)(out + off) = v;
            }
        }
    }
}

int main() {
    const int M = 131072;
    for (int C : {320, 640, 1280}) {
        const int Mc = M * 320 / C;
        size_t bytes = (size_t)Mc * C * 2;
        unsigned short *res, *out;
        hipMalloc(&res, bytes * 4); hipMalloc(&out, bytes * 4);  // 4 rotating copies: no cache reuse between launches
        hipMemset(res, 1, bytes * 4); hipMemset(out, 0, bytes * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto run = [&](auto kern, const char* name, int nbytes_mult) {
            const int tiles = Mc / 256;
            for (int grid : {256, tiles}) {
                for (int i = 0; i < 4; ++i) kern<<<grid, 512>>>(res + (size_t)(i % 4) * bytes / 2, out + (size_t)(i % 4) * bytes / 2, Mc, C, tiles);
                hipEventRecord(e0);
                const int it = 40;
                for (int i = 0; i < it; ++i) kern<<<grid, 512>>>(res + (size_t)(i % 4) * bytes / 2, out + (size_t)(i % 4) * bytes / 2, Mc, C, tiles);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                printf("C=%4d %-22s grid=%5d  %7.1f us  %7.1f GB/s\n", C, name, grid, ms / it * 1e3, (double)bytes * nbytes_mult / (ms / it * 1e-3) / 1e9);
            }
        };
        run(k<0, false>, "epilogue-shape store", 1);
        run(k<1, false>, "whole-row store", 1);
        run(k<0, true>, "epilogue-shape +res", 2);
        run(k<1, true>, "whole-row +res", 2);
        hipFree(res); hipFree(out);
    }
    return 0;
}
