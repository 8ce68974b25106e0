/* hardware.c -- host CPU feature record.
 *
 * Replaces pll_hardware / pll_hardware_probe / pll_hardware_dump /
 * pll_hardware_ignore (hardware.c:159-189 of the reference).  The reference
 * uses the record to pick an x86 kernel variant; nothing here depends on it (all
 * likelihood arithmetic runs on the GPU), but clients call the probe and print
 * the record, so the symbols and the struct layout (pll.h:181-199) are kept.
 */
#include <stdio.h>

#include "internal.h"

pll_hardware_t pll_hardware = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

int pll_hardware_probe(void)
{
#if defined(__x86_64__) || defined(__i386__)
  __builtin_cpu_init();
  pll_hardware.mmx_present = __builtin_cpu_supports("mmx") ? 1 : 0;
  pll_hardware.sse_present = __builtin_cpu_supports("sse") ? 1 : 0;
  pll_hardware.sse2_present = __builtin_cpu_supports("sse2") ? 1 : 0;
  pll_hardware.sse3_present = __builtin_cpu_supports("sse3") ? 1 : 0;
  pll_hardware.ssse3_present = __builtin_cpu_supports("ssse3") ? 1 : 0;
  pll_hardware.sse41_present = __builtin_cpu_supports("sse4.1") ? 1 : 0;
  pll_hardware.sse42_present = __builtin_cpu_supports("sse4.2") ? 1 : 0;
  pll_hardware.popcnt_present = __builtin_cpu_supports("popcnt") ? 1 : 0;
  pll_hardware.avx_present = __builtin_cpu_supports("avx") ? 1 : 0;
  pll_hardware.avx2_present = __builtin_cpu_supports("avx2") ? 1 : 0;
#endif
  pll_hardware.init = 1;
  return PLL_SUCCESS;
}

void pll_hardware_dump(void)
{
  if (!pll_hardware.init) pll_hardware_probe();
  fprintf(stderr, "host CPU features:");
  if (pll_hardware.mmx_present) fprintf(stderr, " mmx");
  if (pll_hardware.sse_present) fprintf(stderr, " sse");
  if (pll_hardware.sse2_present) fprintf(stderr, " sse2");
  if (pll_hardware.sse3_present) fprintf(stderr, " sse3");
  if (pll_hardware.ssse3_present) fprintf(stderr, " ssse3");
  if (pll_hardware.sse41_present) fprintf(stderr, " sse4.1");
  if (pll_hardware.sse42_present) fprintf(stderr, " sse4.2");
  if (pll_hardware.popcnt_present) fprintf(stderr, " popcnt");
  if (pll_hardware.avx_present) fprintf(stderr, " avx");
  if (pll_hardware.avx2_present) fprintf(stderr, " avx2");
  fprintf(stderr, " (unused: likelihood kernels run on %d HIP device(s))\n", pll_amd_device_count());
}

void pll_hardware_ignore(void)
{
  /* the reference marks every feature present; same here, same (lack of) effect */
  pll_hardware.init = 1;
  pll_hardware.altivec_present = pll_hardware.mmx_present = pll_hardware.sse_present = 1;
  pll_hardware.sse2_present = pll_hardware.sse3_present = pll_hardware.ssse3_present = 1;
  pll_hardware.sse41_present = pll_hardware.sse42_present = pll_hardware.popcnt_present = 1;
  pll_hardware.avx_present = pll_hardware.avx2_present = 1;
}
