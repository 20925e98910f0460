/* the reference filter includes "transform360/..." (README step 8 rewrites the path for an in-tree build);
 * the installed headers live under include/Transform360 */
#include "Transform360/VideoFrameTransformHandler.h"
