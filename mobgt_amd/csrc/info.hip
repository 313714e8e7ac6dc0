#include "mobgt_hip.h"
extern "C" int mobgt_abi_version(void) { return 1; }
extern "C" const char* mobgt_build_info(void) { return "libmobgt_hip gfx950 (MI355X), wave64, mfma_f32_32x32x16_bf16"; }
