#include "Transform360/VideoFrameTransformHelper.h"
