// Stream-kernel instantiations of unit 4 (stream_variants.inc says which; stream_units.h does the rest).
#define MCPT_STREAM_UNIT 4
#include "stream_units.h"
