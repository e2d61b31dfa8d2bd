// libssrhip.so translation unit: STFT kernels, transform precision float, part 1 (see tu_stft.inc)
#define SSR_TU_T float
#define SSR_TU_PART 1
#include "tu_stft.inc"
