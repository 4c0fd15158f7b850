// Minimal pin of the "16-byte buffer store anomaly" of csrc/pw_wgrad.h.
//
// Hypothesis from the ISA of pw_wgrad_kernel<1,4,1,1> built with one buffer_store_dwordx4 per (row, lane): the compiler places a
// VALU write of the store's DATA registers in the very next issue slot,
//       buffer_store_dwordx4 v[0:3], v18, s[28:31], s2 offen
//       v_or_b32_e32 v0, 2, v70
// The ISA manuals list "VMEM store of more than 64 bits of data followed by a VALU write of the VGPRs holding the write data" as a
// manually-resolved hazard (1 wait state on gfx9, 2 on gfx940+).  LLVM's GCNHazardRecognizer::createsVALUHazard exempts MUBUF stores
// whose soffset is an SGPR ("this adds a cycle"), which covers ONE wait state, not the two this part needs -- so with a register
// soffset nothing is inserted and the last lanes of each 16-lane pass can read the overwritten register.
//
// This file takes the compiler out of the picture: the store and the overwrite are inline assembly on fixed registers, with an
// explicit number of wait states in between, once with an SGPR soffset and once with an immediate 0 (the row offset folded into
// voffset).  Every lane stores four distinct non-zero words and then zeroes the registers; a slot that reads back with a zero in
// it was overwritten before the hardware had read it.
// build + run:  hipcc --offload-arch=gfx950 -O3 -o store_x4_hazard store_x4_hazard.hip && ./store_x4_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned u4 __attribute__((ext_vector_type(4)));

#define STORE_THEN_CLOBBER(SOFF, WAIT)                                                                                           \
    asm volatile("v_mov_b32 v4, %[a]\n v_add_u32 v5, 1, %[a]\n v_add_u32 v6, 2, %[a]\n v_add_u32 v7, 3, %[a]\n s_nop 7\n"              \
                 "buffer_store_dwordx4 v[4:7], %[off], %[rs], " SOFF " offen\n" WAIT                                             \
                 "v_mov_b32 v4, 0\n v_mov_b32 v5, 0\n v_mov_b32 v6, 0\n v_mov_b32 v7, 0\n"                                       \
                 :                                                                                                               \
                 : [a] "v"(val), [off] "v"(voff), [rs] "s"(rs), [so] "s"(so)                                                     \
                 : "v4", "v5", "v6", "v7", "memory")

template <int MODE>
__global__ void __launch_bounds__(256) store_kernel(unsigned* out, unsigned bytes, int rows, int row_bytes) {
    const unsigned long long p = (unsigned long long)out;
    u4 rs;
    rs.x = __builtin_amdgcn_readfirstlane((unsigned)p);
    rs.y = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32) & 0xffffu);
    rs.z = __builtin_amdgcn_readfirstlane(bytes);
    rs.w = 0x00020000u;
    const unsigned gid = blockIdx.x * 256 + threadIdx.x;
    for (int i = 0; i < rows; ++i) {
        const unsigned val = (gid * (unsigned)rows + (unsigned)i) * 4u + 1u;       // word j of slot s holds 4 s + 1 + j: never zero
        const int so = __builtin_amdgcn_readfirstlane(i * row_bytes);
        unsigned voff = gid * 16u;
        if (MODE >= 3) voff += (unsigned)so;                                          // immediate soffset: the row rides in voffset
        if (MODE == 0) STORE_THEN_CLOBBER("%[so]", "");
        if (MODE == 1) STORE_THEN_CLOBBER("%[so]", "s_nop 0\n");
        if (MODE == 2) STORE_THEN_CLOBBER("%[so]", "s_nop 1\n");
        if (MODE == 3) STORE_THEN_CLOBBER("0", "");
        if (MODE == 4) STORE_THEN_CLOBBER("0", "s_nop 0\n");
        if (MODE == 5) STORE_THEN_CLOBBER("0", "s_nop 1\n");
    }
}

int main() {
    const int blocks = 8192, rows = 16;
    const size_t lanes = (size_t)blocks * 256, row_bytes = lanes * 16, bytes = row_bytes * rows;     // 512 MiB
    unsigned* d;
    if (hipMalloc(&d, bytes) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    std::vector<unsigned> h(bytes / 4);
    const char* names[6] = {"sgpr soffset, 0 wait states", "sgpr soffset, 1 wait state ", "sgpr soffset, 2 wait states",
                            "imm  soffset, 0 wait states", "imm  soffset, 1 wait state ", "imm  soffset, 2 wait states"};
    for (int mode = 0; mode < 6; ++mode) {
        long long bad_total = 0;
        long long by_lane16[16] = {0}, by_word[4] = {0};
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(d, 0xff, bytes);
            switch (mode) {
                case 0: store_kernel<0><<<blocks, 256>>>(d, (unsigned)bytes, rows, (int)row_bytes); break;
                case 1: store_kernel<1><<<blocks, 256>>>(d, (unsigned)bytes, rows, (int)row_bytes); break;
                case 2: store_kernel<2><<<blocks, 256>>>(d, (unsigned)bytes, rows, (int)row_bytes); break;
                case 3: store_kernel<3><<<blocks, 256>>>(d, (unsigned)bytes, rows, (int)row_bytes); break;
                case 4: store_kernel<4><<<blocks, 256>>>(d, (unsigned)bytes, rows, (int)row_bytes); break;
                default: store_kernel<5><<<blocks, 256>>>(d, (unsigned)bytes, rows, (int)row_bytes); break;
            }
            if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
            hipMemcpy(h.data(), d, bytes, hipMemcpyDeviceToHost);
            for (int i = 0; i < rows; ++i)
                for (size_t g = 0; g < lanes; ++g) {
                    const unsigned* s = &h[((size_t)i * lanes + g) * 4];
                    const unsigned want = ((unsigned)g * (unsigned)rows + (unsigned)i) * 4u + 1u;
                    bool bad = false;
                    for (int j = 0; j < 4; ++j)
                        if (s[j] != want + j) { bad = true; by_word[j]++; }
                    if (bad) { bad_total++; by_lane16[g & 15]++; }
                }
        }
        printf("%s: %lld bad slots of %lld", names[mode], bad_total, (long long)lanes * rows * 3);
        if (bad_total) {
            printf("   by lane%%16:");
            for (int l = 0; l < 16; ++l) printf(" %lld", by_lane16[l]);
            printf("   by word:");
            for (int j = 0; j < 4; ++j) printf(" %lld", by_word[j]);
        }
        printf("\n");
    }
    hipFree(d);
    return 0;
}
