// Host front end: where renderer configurations (mcsd::Scene) come from and
// where finished frames go.  The reference equivalents are csrt::LoadConfig
// (src/parser/parser.cpp:94-179) and image_io::Write
// (src/utils/image_io.cpp:25-53).
#ifndef MCPT_HOST_FRONTEND_HPP
#define MCPT_HOST_FRONTEND_HPP

#include <string>

#include "mcsd_scene.hpp"

namespace mcpt
{

// Mitsuba-style XML scene -> configuration (xml_scene.cpp).
// `standins`: table of procedural stand-ins for mesh files the scene names but that are not on disk
// (standin_mesh.cpp); empty = a missing file is an error, as in the reference.
mcsd::Scene LoadXmlScene(const std::string &path, const std::string &standins = "");

// Scenes available without files: "cornell-box".
mcsd::Scene BuiltinScene(const std::string &name);

// By suffix: .png (sRGB 8 bit), .exr (float32 scanlines), .pfm, .f32 (raw).
void WriteImage(const std::string &path, const float *frame, int width, int height);

} // namespace mcpt

#endif // MCPT_HOST_FRONTEND_HPP
