// gfx950 / ROCm 7.2: a buffer store of MORE THAN 8 BYTES whose data registers are overwritten by the very next VALU
// instruction loses data when the store's soffset operand is an SGPR.  LLVM's hazard recognizer
// (GCNHazardRecognizer::createsVALUHazard) pads this case only when soffset is NOT a register, so hand-placed code like
// the loop below - and compiler output that happens to re-use the data registers right behind such a store, as
// tdsa_big.hip's column pass did - writes garbage now and then.  With the offset folded into the VGPR address
// (soffset = 0) the compiler inserts its s_nop and the data is always right.
//
//   hipcc --offload-arch=gfx950 -O3 store_hazard.hip -o store_hazard && ./store_hazard
// prints the number of corrupted 16-byte records for both forms (the hazard needs a busy store path: many workgroups).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// FORM 0: store, data registers overwritten by the NEXT instruction; 1: one wait state (s_nop 0) in between;
//      2: two wait states (s_nop 1, what the compiler inserts for the soffset = 0 form); 3: the compiler's own code
template <int FORM>
__global__ void __launch_bounds__(256) k(unsigned* out, int rows, unsigned row_bytes) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out, 0, int(rows * row_bytes), 0x00020000);
  const unsigned t = blockIdx.x * 256 + threadIdx.x, lane_off = t * 16u;
  for (int row = 0; row < rows; ++row) {
    const unsigned soff = unsigned(row) * row_bytes, rw = unsigned(row);
    // record = (row, global thread, row ^ thread, ~row), built in v[20:23] every iteration
#define BUILD "v_mov_b32 v20, %0\n v_mov_b32 v21, %1\n v_xor_b32 v22, %0, %1\n v_not_b32 v23, %0\n s_nop 4\n"
#define STORE "buffer_store_dwordx4 v[20:23], %2, %3, %4 offen\n"
#define CLOBBER "v_mov_b32 v20, 0xdeadbeef\n v_mov_b32 v21, 0xdeadbeef\n v_mov_b32 v22, 0xdeadbeef\n v_mov_b32 v23, 0xdeadbeef\n"
#define OPS ::"v"(rw), "v"(t), "v"(lane_off), "s"(r), "s"(soff) : "v20", "v21", "v22", "v23", "memory"
    if constexpr (FORM == 0) asm volatile(BUILD STORE CLOBBER OPS);
    else if constexpr (FORM == 1) asm volatile(BUILD STORE "s_nop 0\n" CLOBBER OPS);
    else if constexpr (FORM == 2) asm volatile(BUILD STORE "s_nop 1\n" CLOBBER OPS);
    else {
      typedef unsigned u4 __attribute__((ext_vector_type(4)));
      __builtin_amdgcn_raw_buffer_store_b128(u4{rw, t, rw ^ t, ~rw}, r, lane_off + soff, 0, 0);
    }
  }
}

int main() {
  const int blocks = 2048, rows = 64;
  const unsigned row_bytes = blocks * 256 * 16u;          // 8 MiB per row, 512 MiB in all
  unsigned* d;
  if (hipMalloc(&d, size_t(rows) * row_bytes) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
  std::vector<unsigned> h(size_t(rows) * row_bytes / 4);
  const char* names[4] = {"SGPR soffset, data registers overwritten by the next instruction",
                          "SGPR soffset, one wait state (s_nop 0) before the overwrite    ",
                          "SGPR soffset, two wait states (s_nop 1) before the overwrite   ",
                          "offset in the VGPR (soffset = 0), compiler-scheduled builtin   "};
  for (int form = 0; form < 4; ++form) {
    long bad = 0;
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipMemset(d, 0, size_t(rows) * row_bytes);
      if (form == 0) k<0><<<blocks, 256>>>(d, rows, row_bytes);
      else if (form == 1) k<1><<<blocks, 256>>>(d, rows, row_bytes);
      else if (form == 2) k<2><<<blocks, 256>>>(d, rows, row_bytes);
      else k<3><<<blocks, 256>>>(d, rows, row_bytes);
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
      for (int row = 0; row < rows; ++row)
        for (unsigned t = 0; t < unsigned(blocks) * 256; ++t) {
          const unsigned* q = &h[(size_t(row) * row_bytes + size_t(t) * 16) / 4];
          if (q[0] != unsigned(row) || q[1] != t || q[2] != (unsigned(row) ^ t) || q[3] != ~unsigned(row)) ++bad;
        }
    }
    printf("%s: %ld corrupted records of %ld\n", names[form], bad, 3L * rows * blocks * 256);
  }
  return 0;
}
