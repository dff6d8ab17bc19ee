#include "common.h"

extern "C" const char* sputnik_hip_version(void) { return "sputnik_hip 0.1.0 gfx950"; }
