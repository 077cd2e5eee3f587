// TEST INFRASTRUCTURE ONLY -- minimal stand-in for ROS's generated nav_msgs/MapMetaData.h (absent from this image),
// with exactly the fields hector_map_tools/HectorMapTools.h reads: resolution, width, height, origin.position.{x,y}.
// Field names and types follow the message definition (float32 resolution, uint32 width/height, float64 position).
#pragma once
#include <stdint.h>
namespace nav_msgs {
struct MapMetaData {
  float resolution = 0.0f;
  uint32_t width = 0, height = 0;
  struct Pose {
    struct Point { double x = 0.0, y = 0.0, z = 0.0; } position;
    struct Quaternion { double x = 0.0, y = 0.0, z = 0.0, w = 1.0; } orientation;
  } origin;
};
}  // namespace nav_msgs
