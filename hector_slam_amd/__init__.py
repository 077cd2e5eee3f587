"""hector_slam_amd: MI355X-native scan-to-map Gauss-Newton matcher for hector_mapping."""
