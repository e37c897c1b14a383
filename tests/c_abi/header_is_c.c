/* include/shg_hip.h must be usable from plain C (the drop-in boundary has no C++ or torch types). */
#include "shg_hip.h"

int header_is_c(void) {
    return SHG_ABI_VERSION;
}
