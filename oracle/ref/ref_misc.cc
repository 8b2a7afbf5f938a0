// oracle/_ref: small dependency-free pieces of the reference compiled as they are.
//   ORB_SLAM2::LineIterator   src/lineIterator.cpp:34-77 / include/lineIterator.h  (the Bresenham walk behind
//                             Frame::AssignFeaturesToGridForLine, src/Frame.cc:295-320)
// TEST INFRASTRUCTURE ONLY.
#include <cstdint>
#include <utility>

#include "lineIterator.h"

extern "C" {

// cells visited by the reference's iterator for the segment (x1, y1)-(x2, y2); returns their number (may exceed cap)
int ref_line_iterator(double x1, double y1, double x2, double y2, int32_t* xy, int cap) {
  ORB_SLAM2::LineIterator it(x1, y1, x2, y2);
  std::pair<int, int> p;
  int n = 0;
  while (it.getNext(p)) {
    if (n < cap) { xy[2 * n] = p.first; xy[2 * n + 1] = p.second; }
    n++;
  }
  return n;
}

}  // extern "C"
