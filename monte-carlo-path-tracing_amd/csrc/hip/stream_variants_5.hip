// Stream-kernel instantiations of unit 5 (stream_variants.inc says which; stream_units.h does the rest).
#define MCPT_STREAM_UNIT 5
#include "stream_units.h"
