"""consenrich_amd -- MI355X (gfx950) implementation of Consenrich's estimator hot path.

Modules
  cconsenrich : drop-in callables for the reference's ``consenrich.cconsenrich`` hot-path functions
  batch       : device-resident multi-chain batch (chromosomes per launch), used by the genome driver / bench
  sharding    : longest-processing-time contig sharding across the GPUs of a node + the final track gather
  build       : hipcc build of csrc/ -> lib/libconsenrich_amd.so (C ABI in include/consenrich_amd.h)

There is no CPU compute path in this package.
"""
__version__ = "0.1.0"
