// Microbenchmark: issue cost of the vector instructions the traversal kernel is made of, on gfx950. Every kernel
// runs a long unrolled stream of one instruction kind over eight independent register sets; the grid puts W
// wavefronts on every SIMD of the chip, and the figure printed is SIMD cycles per wave-instruction at that occupancy
// (clock taken from s_memtime-free timing: elapsed time x the device's clock rate).
//   hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float __attribute__((ext_vector_type(2))) Float2;

// 8 instructions per REP8, 64 per loop iteration
#define REP8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
#define BODY(I) REP8(I) REP8(I) REP8(I) REP8(I) REP8(I) REP8(I) REP8(I) REP8(I)

#define KERNEL(NAME, DECL, INSTR, SINK)                                                         \
    __global__ void __launch_bounds__(256) NAME(float *out, int iters) {                       \
        DECL                                                                                    \
        for (int it = 0; it < iters; ++it) { BODY(INSTR) }                                      \
        SINK                                                                                    \
    }

#define DECL_F float a[8], b = threadIdx.x * 1e-3f + 1.0f, c = 0.25f; for (int k = 0; k < 8; ++k) a[k] = threadIdx.x + k;
#define SINK_F float s = 0; for (int k = 0; k < 8; ++k) s += a[k]; out[blockIdx.x * blockDim.x + threadIdx.x] = s;
#define DECL_U uint32_t a[8], b = threadIdx.x * 2654435761u + 1u, c = 0x03020100u; for (int k = 0; k < 8; ++k) a[k] = threadIdx.x * 7u + k;
#define SINK_U uint32_t s = 0; for (int k = 0; k < 8; ++k) s += a[k]; out[blockIdx.x * blockDim.x + threadIdx.x] = (float) s;
#define DECL_P Float2 a[8], b = { threadIdx.x * 1e-3f + 1.0f, 1.5f }, c = { 0.25f, 0.5f }; for (int k = 0; k < 8; ++k) a[k] = (Float2) { (float) threadIdx.x, (float) k };
#define SINK_P float s = 0; for (int k = 0; k < 8; ++k) s += a[k].x + a[k].y; out[blockIdx.x * blockDim.x + threadIdx.x] = s;

#define I_FMA(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define I_PKFMA(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define I_PKFMA_SEL(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(a[k]) : "v"(b), "v"(c));
#define I_PKMUL(k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define I_MAX3(k) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define I_MAX(k) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define I_CVTUB(k) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a[k]) : "v"(b));
#define I_RCP(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
#define I_MULLO(k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define I_MUL24(k) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define I_CMPADDC(k) asm volatile("v_cmp_le_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(a[k]) : "v"(__uint_as_float(b)), "v"(__uint_as_float(c)) : "vcc");
#define I_CMPCND(k) asm volatile("v_cmp_le_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(__uint_as_float(b)), "v"(__uint_as_float(c)) : "vcc");
#define I_PERM(k) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define I_BITOP3(k) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xc8" : "+v"(a[k]) : "v"(b), "v"(c));
#define I_LSHLV(k) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(a[k]) : "v"(c));
#define I_ANDOR(k) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define I_BFE(k) asm volatile("v_bfe_u32 %0, %0, 5, 3" : "+v"(a[k]));
#define I_BCNT(k) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define I_FFBL(k) asm volatile("v_ffbl_b32 %0, %0" : "+v"(a[k]));
#define I_MOV(k) asm volatile("v_mov_b32 %0, %1" : "=v"(a[k]) : "v"(b));
#define I_MBCNT(k) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, %0" : "+v"(a[k]));
#define I_SDWA(k) asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(a[k]) : "v"(c));
#define I_DOT4(k) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
#define I_ALIGNBIT(k) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a[k]) : "v"(b));
#define I_MED3(k) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define I_DIVSCALE(k) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a[k]) : "v"(b) : "vcc");
#define I_DIVFIXUP(k) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define I_MAD64(k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(p[k]) : "v"(b), "v"(c) : "vcc");
#define DECL_Q unsigned long long p[8]; uint32_t b = threadIdx.x * 2654435761u + 1u, c = 0x03020100u; for (int k = 0; k < 8; ++k) p[k] = threadIdx.x * 7u + k;
#define SINK_Q unsigned long long s = 0; for (int k = 0; k < 8; ++k) s += p[k]; out[blockIdx.x * blockDim.x + threadIdx.x] = (float) s;
#define I_LSHLADD64(k) asm volatile("v_lshl_add_u64 %0, %0, 4, %0" : "+v"(p[k]));

KERNEL(k_fma, DECL_F, I_FMA, SINK_F)
KERNEL(k_pkfma, DECL_P, I_PKFMA, SINK_P)
KERNEL(k_pkfma_sel, DECL_P, I_PKFMA_SEL, SINK_P)
KERNEL(k_pkmul, DECL_P, I_PKMUL, SINK_P)
KERNEL(k_max3, DECL_F, I_MAX3, SINK_F)
KERNEL(k_max, DECL_F, I_MAX, SINK_F)
KERNEL(k_med3, DECL_F, I_MED3, SINK_F)
KERNEL(k_cvtub, DECL_F, I_CVTUB, SINK_F)
KERNEL(k_rcp, DECL_F, I_RCP, SINK_F)
KERNEL(k_divscale, DECL_F, I_DIVSCALE, SINK_F)
KERNEL(k_divfixup, DECL_F, I_DIVFIXUP, SINK_F)
KERNEL(k_mullo, DECL_U, I_MULLO, SINK_U)
KERNEL(k_mul24, DECL_U, I_MUL24, SINK_U)
KERNEL(k_cmp_addc, DECL_U, I_CMPADDC, SINK_U)
KERNEL(k_cmp_cndmask, DECL_U, I_CMPCND, SINK_U)
KERNEL(k_perm, DECL_U, I_PERM, SINK_U)
KERNEL(k_bitop3, DECL_U, I_BITOP3, SINK_U)
KERNEL(k_lshl_var, DECL_U, I_LSHLV, SINK_U)
KERNEL(k_and_or, DECL_U, I_ANDOR, SINK_U)
KERNEL(k_bfe, DECL_U, I_BFE, SINK_U)
KERNEL(k_bcnt, DECL_U, I_BCNT, SINK_U)
KERNEL(k_ffbl, DECL_U, I_FFBL, SINK_U)
KERNEL(k_mov, DECL_U, I_MOV, SINK_U)
KERNEL(k_mbcnt, DECL_U, I_MBCNT, SINK_U)
KERNEL(k_sdwa_shift, DECL_U, I_SDWA, SINK_U)
KERNEL(k_dot4, DECL_U, I_DOT4, SINK_U)
KERNEL(k_alignbit, DECL_U, I_ALIGNBIT, SINK_U)
KERNEL(k_mad64, DECL_Q, I_MAD64, SINK_Q)
KERNEL(k_lshladd64, DECL_Q, I_LSHLADD64, SINK_Q)


#define I_CMP(k) asm volatile("v_cmp_le_f32 vcc, %0, %1" : : "v"(__uint_as_float(a[k])), "v"(__uint_as_float(b)) : "vcc");
#define I_CMP64(k) asm volatile("v_cmp_le_f32_e64 s[20:21], %0, %1" : : "v"(__uint_as_float(a[k])), "v"(__uint_as_float(b)) : "s20", "s21");
#define I_CND32(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b) : "vcc");
#define I_CND64(k) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[k]) : "v"(b) : "s20", "s21");
#define I_ADDC(k) asm volatile("v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(a[k]) : : "vcc");
#define I_ADDC64(k) asm volatile("v_addc_co_u32_e64 %0, s[22:23], %0, %0, s[20:21]" : "+v"(a[k]) : : "s20", "s21", "s22", "s23");
#define I_MULF(k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define I_ADDF(k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define I_SUBF(k) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define I_MINF(k) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define I_FMAC(k) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define I_FMAC_SDWA(k) asm volatile("v_fmac_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(a[k]) : "v"(b), "v"(c));
#define I_MULF_SDWA(k) asm volatile("v_mul_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(a[k]) : "v"(b), "v"(c));
#define I_LDEXP(k) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
#define I_AND(k) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define I_OR(k) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define I_XOR(k) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define I_OR3(k) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define I_LSHLOR(k) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a[k]) : "v"(b));
#define I_LSHLADD(k) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[k]) : "v"(b));
#define I_ADDU(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define I_LSHR(k) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a[k]));
#define I_PKMOV_Q(k) asm volatile("v_pk_mov_b32 %0, %1, %1" : "=v"(p[k]) : "v"(p[(k + 1) & 7]));
#define I_MOV64_Q(k) asm volatile("v_mov_b64 %0, %1" : "=v"(p[k]) : "v"(p[(k + 1) & 7]));
#define I_PKFMA16(k) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define I_PKMAX16(k) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define I_PKMAX316(k) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define I_PKADD16(k) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define I_MAXIMUM3(k) asm volatile("v_maximum3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define I_FMA_S(k) asm volatile("v_fma_f32 %0, %0, s20, %1" : "+v"(a[k]) : "v"(c) : "s20");
#define I_CVTUB0(k) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(a[k]) : "v"(b));
#define I_CVTUB3(k) asm volatile("v_cvt_f32_ubyte3 %0, %1" : "=v"(a[k]) : "v"(b));
#define I_CVT_U32(k) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(a[k]) : "v"(b));
#define I_CVT_SDWA(k) asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(a[k]) : "v"(b));
#define I_VALU_SALU(k) asm volatile("v_and_b32 %0, %0, %1\n\ts_and_b64 s[20:21], s[20:21], s[22:23]" : "+v"(a[k]) : "v"(b) : "s20", "s21", "scc");
#define I_DSREAD(k) asm volatile("ds_read_u8 %0, %1" : "=v"(a[k]) : "v"(b & 0x7ffu)); 
#define DECL_D uint32_t a[8], b = (threadIdx.x * 2654435761u >> 7) & 0x7ffu, c = 0; for (int k = 0; k < 8; ++k) a[k] = threadIdx.x * 7u + k;
#define SINK_D asm volatile("s_waitcnt lgkmcnt(0)"); uint32_t s = 0; for (int k = 0; k < 8; ++k) s += a[k]; out[blockIdx.x * blockDim.x + threadIdx.x] = (float) s;

KERNEL(k_cmp, DECL_U, I_CMP, SINK_U)
KERNEL(k_cmp64, DECL_U, I_CMP64, SINK_U)
KERNEL(k_cnd32, DECL_U, I_CND32, SINK_U)
KERNEL(k_cnd64, DECL_U, I_CND64, SINK_U)
KERNEL(k_addc, DECL_U, I_ADDC, SINK_U)
KERNEL(k_addc64, DECL_U, I_ADDC64, SINK_U)
KERNEL(k_mulf, DECL_F, I_MULF, SINK_F)
KERNEL(k_addf, DECL_F, I_ADDF, SINK_F)
KERNEL(k_subf, DECL_F, I_SUBF, SINK_F)
KERNEL(k_minf, DECL_F, I_MINF, SINK_F)
KERNEL(k_fmac, DECL_F, I_FMAC, SINK_F)
KERNEL(k_mulf_sdwa, DECL_F, I_MULF_SDWA, SINK_F)
KERNEL(k_ldexp, DECL_F, I_LDEXP, SINK_F)
KERNEL(k_and, DECL_U, I_AND, SINK_U)
KERNEL(k_or, DECL_U, I_OR, SINK_U)
KERNEL(k_xor, DECL_U, I_XOR, SINK_U)
KERNEL(k_or3, DECL_U, I_OR3, SINK_U)
KERNEL(k_lshlor, DECL_U, I_LSHLOR, SINK_U)
KERNEL(k_lshladd, DECL_U, I_LSHLADD, SINK_U)
KERNEL(k_addu, DECL_U, I_ADDU, SINK_U)
KERNEL(k_lshr, DECL_U, I_LSHR, SINK_U)
KERNEL(k_pkmov, DECL_Q, I_PKMOV_Q, SINK_Q)
KERNEL(k_mov64, DECL_Q, I_MOV64_Q, SINK_Q)
KERNEL(k_pkfma16, DECL_U, I_PKFMA16, SINK_U)
KERNEL(k_pkmax16, DECL_U, I_PKMAX16, SINK_U)
KERNEL(k_pkmax316, DECL_U, I_PKMAX316, SINK_U)
KERNEL(k_pkadd16, DECL_U, I_PKADD16, SINK_U)
KERNEL(k_maximum3, DECL_F, I_MAXIMUM3, SINK_F)
KERNEL(k_fma_s, DECL_F, I_FMA_S, SINK_F)
KERNEL(k_cvtub0, DECL_F, I_CVTUB0, SINK_F)
KERNEL(k_cvtub3, DECL_F, I_CVTUB3, SINK_F)
KERNEL(k_cvtu32, DECL_F, I_CVT_U32, SINK_F)
KERNEL(k_cvt_sdwa, DECL_F, I_CVT_SDWA, SINK_F)
KERNEL(k_valu_salu, DECL_U, I_VALU_SALU, SINK_U)
KERNEL(k_dsread, DECL_D, I_DSREAD, SINK_D)

struct Entry { const char *name; void (*fn)(float *, int); int perSlot; };

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double clockHz = prop.clockRate * 1e3;
    printf("%s: %d CUs, %.0f MHz\n", prop.name, cus, clockHz / 1e6);
    float *out;
    CHECK(hipMalloc(&out, (size_t) cus * 8 * 256 * sizeof(float)));
    const Entry entries[] = {
        { "v_fma_f32", k_fma, 1 }, { "v_pk_fma_f32", k_pkfma, 1 }, { "v_pk_fma_f32 op_sel_hi", k_pkfma_sel, 1 }, { "v_pk_mul_f32", k_pkmul, 1 },
        { "v_max3_f32", k_max3, 1 }, { "v_max_f32", k_max, 1 }, { "v_med3_f32", k_med3, 1 }, { "v_cvt_f32_ubyte1", k_cvtub, 1 }, { "v_rcp_f32", k_rcp, 1 },
        { "v_div_scale_f32", k_divscale, 1 }, { "v_div_fixup_f32", k_divfixup, 1 },
        { "v_mul_lo_u32", k_mullo, 1 }, { "v_mul_u32_u24", k_mul24, 1 }, { "v_cmp_le_f32 + v_addc_co_u32 (pair)", k_cmp_addc, 1 },
        { "v_cmp_le_f32 + v_cndmask_b32 (pair)", k_cmp_cndmask, 1 }, { "v_perm_b32", k_perm, 1 }, { "v_bitop3_b32", k_bitop3, 1 },
        { "v_lshlrev_b32 (vector amount)", k_lshl_var, 1 }, { "v_and_or_b32", k_and_or, 1 }, { "v_bfe_u32", k_bfe, 1 }, { "v_bcnt_u32_b32", k_bcnt, 1 },
        { "v_ffbl_b32", k_ffbl, 1 }, { "v_mov_b32", k_mov, 1 }, { "v_mbcnt_lo", k_mbcnt, 1 }, { "v_lshlrev_b32_sdwa", k_sdwa_shift, 1 },
        { "v_dot4_u32_u8", k_dot4, 1 }, { "v_alignbit_b32", k_alignbit, 1 }, { "v_mad_u64_u32", k_mad64, 1 }, { "v_lshl_add_u64", k_lshladd64, 1 },
        { "v_cmp_le_f32 (vcc)", k_cmp, 1 }, { "v_cmp_le_f32_e64 (sgpr pair)", k_cmp64, 1 }, { "v_cndmask_b32 (vcc)", k_cnd32, 1 }, { "v_cndmask_b32_e64 (sgpr pair)", k_cnd64, 1 },
        { "v_addc_co_u32 (vcc)", k_addc, 1 }, { "v_addc_co_u32_e64 (sgpr pairs)", k_addc64, 1 }, { "v_mul_f32", k_mulf, 1 }, { "v_add_f32", k_addf, 1 }, { "v_sub_f32", k_subf, 1 },
        { "v_ldexp_f32", k_ldexp, 1 }, { "v_and_b32", k_and, 1 }, { "v_or_b32", k_or, 1 }, { "v_xor_b32", k_xor, 1 }, { "v_or3_b32", k_or3, 1 }, { "v_lshl_or_b32", k_lshlor, 1 },
        { "v_lshl_add_u32", k_lshladd, 1 }, { "v_add_u32", k_addu, 1 }, { "v_lshrrev_b32 (constant)", k_lshr, 1 }, { "v_pk_mov_b32", k_pkmov, 1 }, { "v_mov_b64", k_mov64, 1 },
        { "v_pk_fma_f16", k_pkfma16, 1 }, { "v_pk_max_f16", k_pkmax16, 1 }, { "v_pk_maximum3_f16", k_pkmax316, 1 }, { "v_pk_add_f16", k_pkadd16, 1 }, { "v_maximum3_f32", k_maximum3, 1 },
        { "v_fma_f32 (one scalar operand)", k_fma_s, 1 }, { "v_cvt_f32_ubyte0", k_cvtub0, 1 }, { "v_cvt_f32_ubyte3", k_cvtub3, 1 }, { "v_cvt_f32_u32", k_cvtu32, 1 },
        { "v_cvt_f32_u32_sdwa (byte source)", k_cvt_sdwa, 1 }, { "v_and_b32 + s_and_b64 (pair)", k_valu_salu, 1 }, { "ds_read_u8 (random byte of 2 KB)", k_dsread, 1 },
    };
    const int iters = 4096;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%-40s %10s %10s %10s %10s\n", "SIMD cycles per wave-instruction at", "1 wave", "2 waves", "4 waves", "8 waves/SIMD");
    for (const Entry &e : entries) {
        printf("%-40s", e.name);
        for (int wavesPerSimd : { 1, 2, 4, 8 }) {
            // a 256-thread workgroup puts one wave on each SIMD of a CU
            const int blocks = cus * wavesPerSimd;
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, 64);        // warm-up
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, iters);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double instrPerSimd = (double) iters * 64.0 * wavesPerSimd;
            printf(" %10.2f", ms * 1e-3 * clockHz / instrPerSimd);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
