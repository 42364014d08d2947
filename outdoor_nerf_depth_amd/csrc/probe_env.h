// Experiment switches (kernel selection, component removal, stream placement) exist only in diagnostic builds:
// compiled with -DNERFPP_PROBES (tools/probes/build_probes.sh -> csrc/build/variants/lib*_probes.so) they read the
// environment; the shipped libraries read nothing from it, so no environment variable can change what the product
// path computes.
#pragma once
#include <stdlib.h>
#ifdef NERFPP_PROBES
#define PROBE_GETENV(name) getenv(name)
#else
#define PROBE_GETENV(name) ((const char*)nullptr)
#endif
