"""TEST INFRASTRUCTURE ONLY: CPU checkers for the hector scan-match path (see hector_oracle.cpp)."""
