// libssrhip.so translation unit: STFT kernels, transform precision double, part 1 (see tu_stft.inc)
#define SSR_TU_T double
#define SSR_TU_PART 1
#include "tu_stft.inc"
