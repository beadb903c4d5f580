// Mitsuba-style XML scene front end (placeholder until the parser lands).
#include <stdexcept>

#include "frontend.hpp"

namespace mcpt
{

mcsd::Scene LoadXmlScene(const std::string &path)
{
    throw std::runtime_error("XML scene loading is not available in this build ('" + path + "').");
}

} // namespace mcpt
