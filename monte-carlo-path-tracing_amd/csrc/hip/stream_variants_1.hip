// Stream-kernel instantiations of unit 1 (stream_variants.inc says which; stream_units.h does the rest).
#define MCPT_STREAM_UNIT 1
#include "stream_units.h"
