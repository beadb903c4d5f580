// Stream-kernel instantiations of unit 7 (stream_variants.inc says which; stream_units.h does the rest).
#define MCPT_STREAM_UNIT 7
#include "stream_units.h"
