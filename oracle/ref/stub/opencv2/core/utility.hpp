// see opencv2/core/core.hpp in this directory tree: one stand-in header serves all OpenCV includes of the reference
#include "core.hpp"
