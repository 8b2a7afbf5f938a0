#pragma once
#include <opencv2/core/core.hpp>
namespace cv { namespace line_descriptor {
struct KeyLine {   // field order per descriptor_custom.hpp:105-175
  float angle; int class_id; int octave; Point2f pt; float response; float size;
  float startPointX, startPointY, endPointX, endPointY, sPointInOctaveX, sPointInOctaveY, ePointInOctaveX, ePointInOctaveY;
  float lineLength; int numOfPixels;
};
} }
