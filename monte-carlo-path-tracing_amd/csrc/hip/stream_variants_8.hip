// Stream-kernel instantiations of unit 8 (stream_variants.inc says which; stream_units.h does the rest).
#define MCPT_STREAM_UNIT 8
#include "stream_units.h"
