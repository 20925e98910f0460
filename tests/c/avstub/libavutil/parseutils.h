/* tests/c/avstub: see ../avfilter.h */
#include "../avfilter.h"
