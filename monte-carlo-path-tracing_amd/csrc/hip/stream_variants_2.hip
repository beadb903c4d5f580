// Stream-kernel instantiations of unit 2 (stream_variants.inc says which; stream_units.h does the rest).
#define MCPT_STREAM_UNIT 2
#include "stream_units.h"
