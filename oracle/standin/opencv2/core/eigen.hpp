// oracle/standin/opencv2/core/eigen.hpp -- intentionally empty.  Scancontext.h includes this header but
// Scancontext.cpp uses nothing from it; the stub only lets the unmodified reference source compile
// in an image without OpenCV / PCL / ROS (oracle/ref_sc.cpp).  TEST INFRASTRUCTURE ONLY.
#pragma once
