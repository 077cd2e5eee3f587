"""bench.py's legs: common (inputs, the driver's line), cpu (reference CPU baselines), pmc (rocprofv3 counter passes, roofline blocks),
headline (BASELINE configs[2], N >= 1), extra (the other BASELINE configs), group (the single-process C++ deployment shape).
The driver's entry point stays bench.py at the repository root."""
