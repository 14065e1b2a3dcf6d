// Host-side scene model: the part of the reference that STAYS on the host (plugin surface).
// Mirrors the reference's public types so that code written against
//   SourceCode/include/tracer/{Vector,Camera,Material,Texture,Scene}.h
// reads the same here: Scene{sceneSettings, camera, textures, materials, lights, objects},
// Mesh{material, vertices, triangles}, Light{position, intentsity}, Camera{truck,pan,tilt,roll}.
// Differences, all deliberate:
//  * textures are a RUNTIME property of a scene (the reference switches them with the
//    compile-time macro USE_TEXTURES, CMakeLists.txt:18-19, which changes struct layouts);
//    a material whose `texture` is -1 shades with its constant albedo exactly like the
//    non-texture build does (RayTracer.cpp:329);
//  * Mesh refers to its material by index instead of `const Material&` (Scene.h:27), so scenes are
//    freely copyable / movable;
//  * no virtual Texture hierarchy: a texture is a tagged record the device can read.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace crt {

struct Vector {
  float x = 0.0f, y = 0.0f, z = 0.0f;
  Vector() = default;
  Vector(float x, float y, float z) : x(x), y(y), z(z) {}
  float &operator[](unsigned short i) { return i == 0 ? x : (i == 1 ? y : z); }
  const float &operator[](unsigned short i) const { return i == 0 ? x : (i == 1 ? y : z); }
  Vector operator-(const Vector &o) const { return {x - o.x, y - o.y, z - o.z}; }
  Vector operator+(const Vector &o) const { return {x + o.x, y + o.y, z + o.z}; }
  Vector &operator+=(const Vector &o) { x += o.x; y += o.y; z += o.z; return *this; }
  float dot(const Vector &o) const { return x * o.x + y * o.y + z * o.z; }
  Vector cross(const Vector &o) const { return {y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x}; }
  Vector operator*(float s) const { return {x * s, y * s, z * s}; }
  float length() const;
  void normalize();
};
typedef Vector Color;
typedef Vector Albedo;

// 3x3 row-major rotation, applied as row-vector x matrix (reference: Matrix.h:137-142)
struct Matrix3 {
  float m[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  static Matrix3 identity() { return Matrix3(); }
  Matrix3 operator*(const Matrix3 &r) const;  // Matrix.h:122-135
};
Vector operator*(const Vector &v, const Matrix3 &m);

// reference: Camera.h:5-21, Camera.cpp:33-70 (degreesToRadians uses pi ~ 22/7, Camera.cpp:10-12)
struct Camera {
 private:
  Vector position;
  Matrix3 rotationMatrix;

 public:
  Camera() = default;
  explicit Camera(const Vector &position) : position(position) {}
  const Vector &getPosition() const { return position; }
  Vector &setPosition() { return position; }
  const Matrix3 &getRotationMatrix() const { return rotationMatrix; }
  Matrix3 &setRotationMatrix() { return rotationMatrix; }
  Camera &truck(const Vector &direction);
  Camera &pan(const float degrees);
  Camera &tilt(const float degrees);
  Camera &roll(const float degrees);
};

enum MaterialType { Diffuse, Reflective, Constant, Refractive };  // Material.h:7

struct Material {  // Material.h:9-30
  Albedo albedo;
  MaterialType type = Diffuse;
  bool smoothShading = false;
  float ior = 1.0f;
  int texture = -1;  // index into Scene::textures, -1 = constant albedo
};

enum TextureKind { AlbedoTexture = 0, EdgeTexture = 1, CheckerTexture = 2, BitmapTexture = 3 };

struct Texture {  // Texture.h:8-59
  std::string name;
  TextureKind kind = AlbedoTexture;
  Color colorA;      // albedo | innerColor | colorA
  Color colorB;      //        | edgeColor  | colorB
  float scalar = 0;  //        | width      | squareSize
  int width = 0, height = 0;
  std::vector<uint8_t> rgb8;  // decoded bitmap, 3 bytes per texel, row 0 = top (what stbi_load returns)
};

struct Image { unsigned int width = 0, height = 0; };
struct SceneSettings {  // Scene.h:14-18
  Color sceneBackgroundColor;
  Image image;
  unsigned int bucketSize = 1;
};
struct Light {  // Scene.h:20-23 (the member name is the reference's spelling)
  Vector position;
  unsigned int intentsity = 0;
};

struct Vertex {  // Vertex.h:4-23
  Vector position, normal, UV;
};
struct Triangle {  // Triangle.h:8-23: three vertex indices into the mesh + the unit face normal
  unsigned int indexes[3];
  Vector normal;
};

class Mesh {  // Scene.h:25-38, Scene.cpp:5-30
 public:
  unsigned int material = 0;
  std::vector<Vertex> vertices;
  std::vector<Triangle> triangles;
  Mesh() = default;
  // computes face normals and the normalised sum of face normals per vertex
  Mesh(unsigned int material, const std::vector<Vertex> &vertices, const std::vector<unsigned int> &indexes);
};

struct Scene {  // Scene.h:50-69
  SceneSettings sceneSettings;
  Camera camera;
  std::vector<Texture> textures;
  std::vector<Material> materials;
  std::vector<Light> lights;
  std::vector<Mesh> objects;
};

}  // namespace crt
