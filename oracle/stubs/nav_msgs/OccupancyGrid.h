// TEST INFRASTRUCTURE ONLY -- minimal stand-in for ROS's generated nav_msgs/OccupancyGrid.h: `info` + the int8 `data`
// vector, and the ConstPtr typedef (boost::shared_ptr in ROS 1; std::shared_ptr has the same interface for what
// HectorMapTools.h does with it: copy, operator->, operator*).
#pragma once
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <memory>
#include <vector>

#include "MapMetaData.h"
namespace nav_msgs {
struct OccupancyGrid {
  MapMetaData info;
  std::vector<int8_t> data;
};
typedef std::shared_ptr<OccupancyGrid> OccupancyGridPtr;
typedef std::shared_ptr<const OccupancyGrid> OccupancyGridConstPtr;
}  // namespace nav_msgs
