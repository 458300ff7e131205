/* lets an unmodified `#include <ebur128.h>` (src/scan.c:31 of loudgain) pick up the
 * HIP-backed subset: add -I<repo>/include/compat to the build */
#include "../loudscan_ebur128.h"
