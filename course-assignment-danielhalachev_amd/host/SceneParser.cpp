#include "SceneParser.h"

#include <zlib.h>

#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <thread>
#include <utility>

namespace crt {
namespace {

// ------------------------------------------------------------------------------------------------
// A small JSON document model.  Arrays made only of numbers (vertices, triangles, colours) are kept
// as a flat vector<double>, which is what makes 200k-triangle scenes cheap to hold.
struct JValue {
  enum Type { Null, Bool, Number, String, Array, Object } type = Null;
  bool boolean = false;
  double number = 0;
  std::string string;
  std::vector<double> numbers;  // Array of numbers only
  std::vector<JValue> items;    // any other Array
  std::vector<std::pair<std::string, JValue>> members;

  const JValue *find(const char *key) const {
    for (auto &m : members)
      if (m.first == key) return &m.second;
    return nullptr;
  }
  bool isNumericArray() const { return type == Array && items.empty(); }
};

class JsonReader {
 public:
  JsonReader(const char *begin, const char *end) : p(begin), end(end), begin(begin) {}
  JValue parseDocument() {
    JValue v = parseValue(0);
    skipSpace();
    if (p != end) fail("trailing characters after the JSON document");
    return v;
  }

 private:
  const char *p, *end, *begin;

  [[noreturn]] void fail(const std::string &what) const {
    throw SceneParseError("JSON error at byte " + std::to_string(p - begin) + ": " + what);
  }
  void skipSpace() {
    while (p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p;
  }
  bool consume(char c) {
    skipSpace();
    if (p < end && *p == c) { ++p; return true; }
    return false;
  }
  void expect(char c) {
    if (!consume(c)) fail(std::string("expected '") + c + "'");
  }
  static void appendUtf8(std::string &s, unsigned cp) {
    if (cp < 0x80) s += (char)cp;
    else if (cp < 0x800) { s += (char)(0xC0 | (cp >> 6)); s += (char)(0x80 | (cp & 0x3F)); }
    else if (cp < 0x10000) { s += (char)(0xE0 | (cp >> 12)); s += (char)(0x80 | ((cp >> 6) & 0x3F)); s += (char)(0x80 | (cp & 0x3F)); }
    else { s += (char)(0xF0 | (cp >> 18)); s += (char)(0x80 | ((cp >> 12) & 0x3F)); s += (char)(0x80 | ((cp >> 6) & 0x3F)); s += (char)(0x80 | (cp & 0x3F)); }
  }
  unsigned hex4() {
    if (end - p < 4) fail("truncated \\u escape");
    unsigned v = 0;
    for (int i = 0; i < 4; i++, ++p) {
      char c = *p;
      v <<= 4;
      if (c >= '0' && c <= '9') v |= (unsigned)(c - '0');
      else if (c >= 'a' && c <= 'f') v |= (unsigned)(c - 'a' + 10);
      else if (c >= 'A' && c <= 'F') v |= (unsigned)(c - 'A' + 10);
      else fail("bad \\u escape");
    }
    return v;
  }
  std::string parseString() {
    expect('"');
    std::string s;
    while (true) {
      if (p >= end) fail("unterminated string");
      char c = *p++;
      if (c == '"') break;
      if (c != '\\') { s += c; continue; }
      if (p >= end) fail("unterminated escape");
      char e = *p++;
      switch (e) {
        case '"': s += '"'; break;
        case '\\': s += '\\'; break;
        case '/': s += '/'; break;
        case 'b': s += '\b'; break;
        case 'f': s += '\f'; break;
        case 'n': s += '\n'; break;
        case 'r': s += '\r'; break;
        case 't': s += '\t'; break;
        case 'u': {
          unsigned cp = hex4();
          if (cp >= 0xD800 && cp < 0xDC00 && end - p >= 6 && p[0] == '\\' && p[1] == 'u') {
            p += 2;
            unsigned lo = hex4();
            cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
          }
          appendUtf8(s, cp);
          break;
        }
        default: fail("bad escape");
      }
    }
    return s;
  }
  double parseNumber() {
    skipSpace();
    const char *s = p;
    if (p < end && (*p == '-' || *p == '+')) ++p;
    bool digits = false;
    while (p < end && ((*p >= '0' && *p <= '9') || *p == '.' || *p == 'e' || *p == 'E' || *p == '-' || *p == '+')) {
      if (*p >= '0' && *p <= '9') digits = true;
      ++p;
    }
    if (!digits) fail("expected a number");
    char buf[64];
    size_t n = (size_t)(p - s);
    if (n < sizeof(buf)) {  // numbers are parsed as binary64 and narrowed later, like RapidJSON's GetFloat
      memcpy(buf, s, n);
      buf[n] = 0;
      return strtod(buf, nullptr);
    }
    return strtod(std::string(s, n).c_str(), nullptr);
  }
  JValue parseValue(int depth) {
    if (depth > 64) fail("nesting too deep");
    skipSpace();
    if (p >= end) fail("unexpected end of input");
    JValue v;
    char c = *p;
    if (c == '{') {
      ++p;
      v.type = JValue::Object;
      if (consume('}')) return v;
      do {
        skipSpace();
        std::string key = parseString();
        expect(':');
        v.members.emplace_back(std::move(key), parseValue(depth + 1));
      } while (consume(','));
      expect('}');
    } else if (c == '[') {
      ++p;
      v.type = JValue::Array;
      if (consume(']')) return v;
      bool numeric = true;
      do {
        skipSpace();
        if (p >= end) fail("unterminated array");
        char f = *p;
        if (numeric && (f == '-' || (f >= '0' && f <= '9'))) {
          v.numbers.push_back(parseNumber());
        } else {
          if (numeric) {  // switch representation: move what was read so far into generic items
            numeric = false;
            for (double d : v.numbers) {
              JValue n;
              n.type = JValue::Number;
              n.number = d;
              v.items.push_back(std::move(n));
            }
            v.numbers.clear();
          }
          v.items.push_back(parseValue(depth + 1));
        }
      } while (consume(','));
      expect(']');
    } else if (c == '"') {
      v.type = JValue::String;
      v.string = parseString();
    } else if (c == 't' && end - p >= 4 && !strncmp(p, "true", 4)) {
      p += 4; v.type = JValue::Bool; v.boolean = true;
    } else if (c == 'f' && end - p >= 5 && !strncmp(p, "false", 5)) {
      p += 5; v.type = JValue::Bool; v.boolean = false;
    } else if (c == 'n' && end - p >= 4 && !strncmp(p, "null", 4)) {
      p += 4; v.type = JValue::Null;
    } else {
      v.type = JValue::Number;
      v.number = parseNumber();
    }
    return v;
  }
};

// ------------------------------------------------------------------------------------------------
const JValue &member(const JValue &obj, const char *key, const char *where) {
  if (obj.type != JValue::Object) throw SceneParseError(std::string(where) + " is not an object");
  const JValue *v = obj.find(key);
  if (!v) throw SceneParseError(std::string("missing key \"") + key + "\" in " + where);
  return *v;
}

// reference: loadFloatSTLVector (SceneParser.cpp:79-86): GetFloat() == static_cast<float>(double)
std::vector<float> floatArray(const JValue &v, size_t expected, const char *what) {
  if (!v.isNumericArray()) throw SceneParseError(std::string(what) + " must be an array of numbers");
  if (expected != 0 && v.numbers.size() != expected)
    throw SceneParseError(std::string(what) + " must have " + std::to_string(expected) + " elements");
  std::vector<float> out(v.numbers.size());
  for (size_t i = 0; i < out.size(); i++) out[i] = static_cast<float>(v.numbers[i]);
  return out;
}
Vector vec3(const JValue &v, const char *what) {
  std::vector<float> f = floatArray(v, 3, what);
  return Vector(f[0], f[1], f[2]);
}
float floatValue(const JValue &v, const char *what) {
  if (v.type != JValue::Number) throw SceneParseError(std::string(what) + " must be a number");
  return static_cast<float>(v.number);
}
unsigned uintValue(const JValue &v, const char *what) {
  if (v.type != JValue::Number || v.number < 0 || v.number > 4294967295.0 || v.number != (double)(uint64_t)v.number)
    throw SceneParseError(std::string(what) + " must be an unsigned integer");
  return (unsigned)v.number;
}

SceneSettings parseSceneSettings(const JValue &doc) {  // SceneParser.cpp:88-114
  SceneSettings s;
  const JValue &settings = member(doc, "settings", "document");
  s.sceneBackgroundColor = vec3(member(settings, "background_color", "settings"), "background_color");
  const JValue &image = member(settings, "image_settings", "settings");
  s.image.width = uintValue(member(image, "width", "image_settings"), "width");
  s.image.height = uintValue(member(image, "height", "image_settings"), "height");
  unsigned hw = std::thread::hardware_concurrency();
  unsigned bucket = (hw == 1) ? 1 : hw * 6;  // SceneParser.cpp:104-105
  const JValue *b = image.type == JValue::Object ? image.find("bucket_size") : nullptr;
  if (b && b->type == JValue::Number && b->number == (double)(int)b->number) bucket = (unsigned)(int)b->number;
  s.bucketSize = bucket;
  return s;
}

Camera parseCameraSettings(const JValue &doc) {  // SceneParser.cpp:116-130
  Camera camera;
  const JValue &cam = member(doc, "camera", "document");
  camera.setPosition() = vec3(member(cam, "position", "camera"), "camera position");
  std::vector<float> m = floatArray(member(cam, "matrix", "camera"), 9, "camera matrix");
  Matrix3 rot;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) rot.m[i][j] = m[3 * i + j];
  camera.setRotationMatrix() = rot;
  return camera;
}

std::vector<Light> parseLightSettings(const JValue &doc) {  // SceneParser.cpp:132-148
  std::vector<Light> lights;
  const JValue *lv = doc.find("lights");
  if (!lv || lv->type != JValue::Array) return lights;
  for (const JValue &l : lv->items) {
    Light light;
    light.intentsity = uintValue(member(l, "intensity", "light"), "light intensity");
    light.position = vec3(member(l, "position", "light"), "light position");
    lights.push_back(light);
  }
  return lights;
}

std::vector<Texture> parseTextures(const JValue &doc, const std::string &sceneFolder) {  // SceneParser.cpp:150-209
  std::vector<Texture> textures;
  const JValue *tv = doc.find("textures");
  if (!tv || tv->type != JValue::Array) return textures;
  for (const JValue &t : tv->items) {
    Texture tex;
    const JValue &type = member(t, "type", "texture");
    const JValue &name = member(t, "name", "texture");
    if (type.type != JValue::String || name.type != JValue::String) throw SceneParseError("texture type/name must be strings");
    tex.name = name.string;
    if (type.string == "albedo") {
      tex.kind = AlbedoTexture;
      tex.colorA = vec3(member(t, "albedo", "albedo texture"), "texture albedo");
    } else if (type.string == "edges") {
      tex.kind = EdgeTexture;
      tex.colorA = vec3(member(t, "inner_color", "edges texture"), "inner_color");
      tex.colorB = vec3(member(t, "edge_color", "edges texture"), "edge_color");
      tex.scalar = floatValue(member(t, "edge_width", "edges texture"), "edge_width");
    } else if (type.string == "checker") {
      tex.kind = CheckerTexture;
      tex.colorA = vec3(member(t, "color_A", "checker texture"), "color_A");
      tex.colorB = vec3(member(t, "color_B", "checker texture"), "color_B");
      tex.scalar = floatValue(member(t, "square_size", "checker texture"), "square_size");
    } else if (type.string == "bitmap") {
      tex.kind = BitmapTexture;
      const JValue &path = member(t, "file_path", "bitmap texture");
      if (path.type != JValue::String) throw SceneParseError("file_path must be a string");
      loadBitmapRGB8(sceneFolder + path.string, tex.width, tex.height, tex.rgb8);  // SceneParser.cpp:201
    } else {
      throw SceneParseError("Invalid material");  // the reference's message, SceneParser.cpp:203
    }
    textures.push_back(std::move(tex));
  }
  return textures;
}

std::vector<Material> parseMaterials(const JValue &doc, const std::vector<Texture> &textures) {  // SceneParser.cpp:211-271
  std::vector<Material> materials;
  const JValue *mv = doc.find("materials");
  if (!mv || mv->type != JValue::Array) return materials;
  for (const JValue &m : mv->items) {
    Material mat;
    const JValue &type = member(m, "type", "material");
    if (type.type != JValue::String) throw SceneParseError("material type must be a string");
    float ior = 0;  // SceneParser.cpp:222
    if (type.string == "diffuse") mat.type = Diffuse;
    else if (type.string == "reflective") mat.type = Reflective;
    else if (type.string == "refractive") {
      mat.type = Refractive;
      ior = floatValue(member(m, "ior", "refractive material"), "ior");
    } else if (type.string == "constant") mat.type = Constant;
    else throw SceneParseError("Invalid material");
    const JValue &smooth = member(m, "smooth_shading", "material");
    if (smooth.type != JValue::Bool) throw SceneParseError("smooth_shading must be a bool");
    mat.smoothShading = smooth.boolean;
    mat.ior = ior;
    mat.albedo = Albedo(0, 0, 0);
    const JValue *albedo = m.find("albedo");
    if (albedo && albedo->type == JValue::String) {
      // textured scenes name a texture here (SceneParser.cpp:242-251): first texture with that name
      mat.texture = -1;
      for (size_t i = 0; i < textures.size(); i++)
        if (textures[i].name == albedo->string) { mat.texture = (int)i; break; }
      if (mat.texture < 0) throw SceneParseError("material refers to unknown texture \"" + albedo->string + "\"");
    } else if (mat.type != Refractive) {  // SceneParser.cpp:260-264: refractive keeps albedo (0,0,0)
      if (!albedo) throw SceneParseError("missing key \"albedo\" in material");
      mat.albedo = vec3(*albedo, "material albedo");
    }
    materials.push_back(mat);
  }
  return materials;
}

std::vector<Mesh> parseSceneObjects(const JValue &doc, const std::vector<Material> &materials) {  // SceneParser.cpp:273-321
  std::vector<Mesh> meshes;
  const JValue *ov = doc.find("objects");
  if (!ov || ov->type != JValue::Array) return meshes;
  meshes.reserve(ov->items.size());
  for (const JValue &o : ov->items) {
    unsigned material = uintValue(member(o, "material_index", "object"), "material_index");
    if (material >= materials.size()) throw SceneParseError("material_index out of range");
    std::vector<float> pos = floatArray(member(o, "vertices", "object"), 0, "vertices");
    if (pos.size() % 3) throw SceneParseError("vertices must hold 3 floats per vertex");
    std::vector<Vertex> vertices(pos.size() / 3);
    for (size_t i = 0; i < vertices.size(); i++) vertices[i].position = Vector(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
    if (const JValue *uv = o.find("uvs")) {
      std::vector<float> uvs = floatArray(*uv, 0, "uvs");
      if (uvs.size() % 3 || uvs.size() / 3 > vertices.size()) throw SceneParseError("uvs must hold 3 floats per vertex");
      for (size_t i = 0; i < uvs.size() / 3; i++) vertices[i].UV = Vector(uvs[3 * i], uvs[3 * i + 1], uvs[3 * i + 2]);
    }
    const JValue &tri = member(o, "triangles", "object");
    if (!tri.isNumericArray() || tri.numbers.size() % 3) throw SceneParseError("triangles must hold 3 indices per triangle");
    std::vector<unsigned int> triples(tri.numbers.size());
    for (size_t i = 0; i < triples.size(); i++) {
      double d = tri.numbers[i];
      if (d < 0 || d >= (double)vertices.size() || d != (double)(uint64_t)d) throw SceneParseError("triangle index out of range");
      triples[i] = (unsigned)d;
    }
    meshes.push_back(Mesh(material, vertices, triples));
  }
  return meshes;
}

}  // namespace

Scene SceneParser::parseSceneText(const std::string &text, const std::string &sceneFolder) {
  JsonReader reader(text.data(), text.data() + text.size());
  JValue doc = reader.parseDocument();
  if (doc.type != JValue::Object) throw SceneParseError("the scene document must be a JSON object");
  Scene scene;  // same order as SceneParser.cpp:46-63
  scene.sceneSettings = parseSceneSettings(doc);
  scene.camera = parseCameraSettings(doc);
  scene.textures = parseTextures(doc, sceneFolder);
  scene.materials = parseMaterials(doc, scene.textures);
  scene.lights = parseLightSettings(doc);
  scene.objects = parseSceneObjects(doc, scene.materials);
  return scene;
}

Scene SceneParser::parseScene(const std::string &pathToScene, const std::string &sceneFolder) {
  const std::string path = (sceneFolder.empty() ? "" : sceneFolder + "/") + pathToScene;  // SceneParser.cpp:40
  std::ifstream ifs(path, std::ios::binary);
  if (!ifs.is_open()) throw SceneParseError("cannot open scene file " + path);
  std::stringstream ss;
  ss << ifs.rdbuf();
  return parseSceneText(ss.str(), sceneFolder);
}

// ------------------------------------------------------------------------------------------------ bitmaps
namespace {

void loadPNM(const std::vector<uint8_t> &d, int &width, int &height, std::vector<uint8_t> &rgb) {
  size_t p = 2;
  auto token = [&]() -> long {
    while (p < d.size()) {
      if (d[p] == '#') { while (p < d.size() && d[p] != '\n') ++p; }
      else if (d[p] == ' ' || d[p] == '\n' || d[p] == '\r' || d[p] == '\t') ++p;
      else break;
    }
    if (p >= d.size() || d[p] < '0' || d[p] > '9') throw SceneParseError("bad PNM header");
    long v = 0;
    while (p < d.size() && d[p] >= '0' && d[p] <= '9') v = v * 10 + (d[p++] - '0');
    return v;
  };
  const char kind = (char)d[1];
  const int channels = (kind == '3' || kind == '6') ? 3 : 1;
  width = (int)token();
  height = (int)token();
  long maxval = token();
  if (width <= 0 || height <= 0 || maxval != 255) throw SceneParseError("unsupported PNM (need 8-bit)");
  const size_t n = (size_t)width * height;
  std::vector<uint8_t> raw(n * channels);
  if (kind == '5' || kind == '6') {
    ++p;  // single whitespace after maxval
    if (d.size() - p < raw.size()) throw SceneParseError("truncated PNM");
    memcpy(raw.data(), d.data() + p, raw.size());
  } else {
    for (size_t i = 0; i < raw.size(); i++) raw[i] = (uint8_t)token();
  }
  rgb.resize(n * 3);
  for (size_t i = 0; i < n; i++)
    for (int c = 0; c < 3; c++) rgb[3 * i + c] = raw[i * channels + (channels == 3 ? c : 0)];
}

uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

void loadPNG(const std::vector<uint8_t> &d, int &width, int &height, std::vector<uint8_t> &rgb) {
  size_t p = 8;
  std::vector<uint8_t> idat, palette;
  int depth = 0, ctype = 0, interlace = 0;
  while (p + 12 <= d.size()) {
    uint32_t len = be32(&d[p]);
    const uint8_t *tag = &d[p + 4];
    if (p + 12 + len > d.size()) throw SceneParseError("truncated PNG");
    const uint8_t *body = &d[p + 8];
    if (!memcmp(tag, "IHDR", 4) && len >= 13) {
      width = (int)be32(body); height = (int)be32(body + 4);
      depth = body[8]; ctype = body[9]; interlace = body[12];
    } else if (!memcmp(tag, "PLTE", 4)) palette.assign(body, body + len);
    else if (!memcmp(tag, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
    else if (!memcmp(tag, "IEND", 4)) break;
    p += 12 + len;
  }
  if (width <= 0 || height <= 0 || depth != 8 || interlace != 0) throw SceneParseError("unsupported PNG (need 8-bit, non-interlaced)");
  int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
  if (!channels) throw SceneParseError("unsupported PNG colour type");
  const size_t stride = (size_t)width * channels;
  std::vector<uint8_t> raw((stride + 1) * height);
  uLongf out_len = (uLongf)raw.size();
  if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size())
    throw SceneParseError("PNG inflate failed");
  std::vector<uint8_t> img(stride * height);
  for (int y = 0; y < height; y++) {
    const uint8_t *src = &raw[(stride + 1) * y];
    uint8_t *dst = &img[stride * y];
    const uint8_t *up = y ? &img[stride * (y - 1)] : nullptr;
    const int filter = src[0];
    for (size_t x = 0; x < stride; x++) {
      int a = x >= (size_t)channels ? dst[x - channels] : 0, b = up ? up[x] : 0, c = (up && x >= (size_t)channels) ? up[x - channels] : 0;
      int v = src[1 + x];
      switch (filter) {
        case 0: break;
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) / 2; break;
        case 4: {
          int pa = abs(b - c), pb = abs(a - c), pc = abs(a + b - 2 * c);
          v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
          break;
        }
        default: throw SceneParseError("bad PNG filter");
      }
      dst[x] = (uint8_t)v;
    }
  }
  const size_t n = (size_t)width * height;
  rgb.resize(n * 3);
  for (size_t i = 0; i < n; i++) {
    const uint8_t *px = &img[i * channels];
    if (ctype == 3) {
      if ((size_t)px[0] * 3 + 2 >= palette.size()) throw SceneParseError("PNG palette index out of range");
      memcpy(&rgb[3 * i], &palette[px[0] * 3], 3);
    } else if (ctype == 0 || ctype == 4) {
      rgb[3 * i] = rgb[3 * i + 1] = rgb[3 * i + 2] = px[0];
    } else {
      memcpy(&rgb[3 * i], px, 3);
    }
  }
}

// Uncompressed Windows BMP, 24 or 32 bits per pixel (BITMAPINFOHEADER and later; bottom-up unless the height is negative), as
// stbi_load returns it: rows top-down, red first; a fourth channel is dropped (Texture.cpp:53-57 reads channels 0..2).
uint32_t le32(const std::vector<uint8_t> &d, size_t p) { return (uint32_t)d[p] | ((uint32_t)d[p + 1] << 8) | ((uint32_t)d[p + 2] << 16) | ((uint32_t)d[p + 3] << 24); }
uint32_t le16(const std::vector<uint8_t> &d, size_t p) { return (uint32_t)d[p] | ((uint32_t)d[p + 1] << 8); }
void loadBMP(const std::vector<uint8_t> &d, int &width, int &height, std::vector<uint8_t> &rgb) {
  if (d.size() < 54) throw SceneParseError("truncated BMP");
  const uint32_t offset = le32(d, 10), header = le32(d, 14);
  const int32_t w = (int32_t)le32(d, 18), hraw = (int32_t)le32(d, 22);
  const uint32_t bpp = le16(d, 28), compression = le32(d, 30);
  if (header < 40 || w <= 0 || hraw == 0 || (bpp != 24 && bpp != 32) || (compression != 0 && !(compression == 3 && bpp == 32)))
    throw SceneParseError("unsupported BMP (need uncompressed 24 or 32 bits per pixel)");
  const int h = hraw < 0 ? -hraw : hraw;
  const size_t stride = (((size_t)w * bpp + 31) / 32) * 4;
  if ((uint64_t)offset + stride * (size_t)h > d.size()) throw SceneParseError("truncated BMP");
  width = w; height = h;
  rgb.resize((size_t)w * h * 3);
  for (int y = 0; y < h; y++) {
    const uint8_t *row = d.data() + offset + stride * (size_t)(hraw < 0 ? y : h - 1 - y);
    for (int x = 0; x < w; x++) {
      const uint8_t *px = row + (size_t)x * (bpp / 8);
      uint8_t *o = &rgb[((size_t)y * w + x) * 3];
      o[0] = px[2]; o[1] = px[1]; o[2] = px[0];   // stored blue first
    }
  }
}

// Truevision TGA, uncompressed true colour (image type 2), 24 or 32 bits per pixel, no colour map; bottom-up unless the
// descriptor's bit 5 says otherwise.  (TGA has no signature: taken for one when the header is consistent and the name ends in .tga.)
void loadTGA(const std::vector<uint8_t> &d, int &width, int &height, std::vector<uint8_t> &rgb) {
  if (d.size() < 18) throw SceneParseError("truncated TGA");
  const uint32_t id_len = d[0], cmap = d[1], type = d[2], w = le16(d, 12), h = le16(d, 14), bpp = d[16], desc = d[17];
  if (cmap != 0 || type != 2 || (bpp != 24 && bpp != 32) || w == 0 || h == 0) throw SceneParseError("unsupported TGA (need uncompressed true colour, 24 or 32 bits per pixel)");
  const size_t start = 18 + id_len, bytes = bpp / 8;
  if (start + (size_t)w * h * bytes > d.size()) throw SceneParseError("truncated TGA");
  width = (int)w; height = (int)h;
  rgb.resize((size_t)w * h * 3);
  for (uint32_t y = 0; y < h; y++) {
    const uint8_t *row = d.data() + start + (size_t)((desc & 0x20u) ? y : h - 1 - y) * w * bytes;
    for (uint32_t x = 0; x < w; x++) {
      const uint8_t *px = row + (size_t)((desc & 0x10u) ? w - 1 - x : x) * bytes;
      uint8_t *o = &rgb[((size_t)y * w + x) * 3];
      o[0] = px[2]; o[1] = px[1]; o[2] = px[0];
    }
  }
}

}  // namespace

void loadBitmapRGB8(const std::string &path, int &width, int &height, std::vector<uint8_t> &rgb8) {
  std::ifstream f(path, std::ios::binary);
  if (!f.is_open()) throw SceneParseError("cannot open bitmap " + path);
  std::vector<uint8_t> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  static const uint8_t png_sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (d.size() >= 8 && !memcmp(d.data(), png_sig, 8)) return loadPNG(d, width, height, rgb8);
  if (d.size() >= 2 && d[0] == 'P' && (d[1] == '2' || d[1] == '3' || d[1] == '5' || d[1] == '6')) return loadPNM(d, width, height, rgb8);
  if (d.size() >= 2 && d[0] == 'B' && d[1] == 'M') return loadBMP(d, width, height, rgb8);
  if (path.size() >= 4 && (path.compare(path.size() - 4, 4, ".tga") == 0 || path.compare(path.size() - 4, 4, ".TGA") == 0)) return loadTGA(d, width, height, rgb8);
  // The reference decodes bitmaps with stb_image (Texture.cpp:46-60: JPEG, GIF, PSD, HDR, 16-bit and interlaced PNG ... as well).  This
  // loader reads 8-bit PNG, PNM, uncompressed BMP and TGA; anything else is decoded by the CALLER and handed over as RGB8:
  // crt_scene_desc::texels / crt_texture (include/crt_hip.h), three lines with stb_image -- INTEGRATION.md, section B.
  throw SceneParseError("unsupported bitmap format (8-bit PNG, PNM, uncompressed BMP or TGA expected; decode anything else with stb_image and "
                        "pass RGB8 texels through crt_scene_desc: INTEGRATION.md section B): " + path);
}

}  // namespace crt
