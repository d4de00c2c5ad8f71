#include "dtc_common.h"
DTC_API const char* dtc_version(void) { return "detectorch_hip 0.1.0"; }
DTC_API const char* dtc_target_arch(void) { return "gfx950"; }
