// `make EXPERIMENTAL=1`: the stream kernel, the queued renderer, the first multi-kernel version and the trace-rate experiment are part
// of this library (hip/formulations_not_built.hip stands in for them otherwise).
namespace mcpt
{
bool FormulationsBuilt() { return true; }
} // namespace mcpt
