// `.crtscene` loader -- the input half of the plugin surface.
// Same schema and conversion rules as the reference's RapidJSON-based loader
// (reference: SourceCode/src/SceneParser.cpp:17-35,39-322; SceneParser.h), re-implemented on a
// small self-contained JSON reader because RapidJSON is not available offline.
#pragma once

#include <stdexcept>
#include <string>

#include "Scene.h"

namespace crt {

struct SceneParseError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

class SceneParser {
 public:
  SceneParser() = default;
  // reference: SceneParser::parseScene(pathToScene, sceneFolder), SceneParser.cpp:39-66.
  // The file read is (sceneFolder.empty() ? "" : sceneFolder + "/") + pathToScene; bitmap textures are
  // loaded from sceneFolder + file_path (no separator, SceneParser.cpp:201).
  Scene parseScene(const std::string &pathToScene, const std::string &sceneFolder = "");
  // same, from JSON text already in memory
  Scene parseSceneText(const std::string &jsonText, const std::string &sceneFolder = "");
};

// Decodes an image file to RGB8 (row 0 = top).  Supported: binary/ASCII PPM (P6/P3), PGM (P5/P2),
// and 8-bit non-interlaced PNG (grey / RGB / RGBA / palette).  Throws SceneParseError otherwise.
void loadBitmapRGB8(const std::string &path, int &width, int &height, std::vector<uint8_t> &rgb8);

}  // namespace crt
