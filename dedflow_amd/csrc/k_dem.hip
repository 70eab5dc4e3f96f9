// DEM contact-force sweep (build-defined: the reference's Particle.c is a storage
// container only).  Filled in below the Krylov/assembly path.
#include "dfl_common.hpp"
