#include "core/core.hpp"
#include "line_descriptor/descriptor.hpp"
