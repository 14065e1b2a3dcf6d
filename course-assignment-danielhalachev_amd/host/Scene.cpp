#include "Scene.h"

#include <cmath>

namespace crt {

float Vector::length() const {  // reference: Vector.cpp:114-117
  return std::sqrt(x * x + y * y + z * z);
}

void Vector::normalize() {  // reference: Vector.cpp:97-106 (multiply by the reciprocal, never rsqrt)
  float length = this->length();
  if (length == 0) return;
  length = 1.0f / length;
  x *= length;
  y *= length;
  z *= length;
}

Matrix3 Matrix3::operator*(const Matrix3 &r) const {  // reference: Matrix.h:122-135 / 144-156
  Matrix3 out;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      out.m[i][j] = 0;
      for (int k = 0; k < 3; k++) out.m[i][j] += m[i][k] * r.m[k][j];
    }
  return out;
}

Vector operator*(const Vector &v, const Matrix3 &r) {  // reference: Matrix.h:137-142
  return Vector(v.x * r.m[0][0] + v.y * r.m[1][0] + v.z * r.m[2][0],
                v.x * r.m[0][1] + v.y * r.m[1][1] + v.z * r.m[2][1],
                v.x * r.m[0][2] + v.y * r.m[1][2] + v.z * r.m[2][2]);
}

static float degreesToRadians(const float degrees) {  // reference: Camera.cpp:10-12 (pi ~ 22/7)
  return degrees * (22 / (7 * 180.0f));
}

Camera &Camera::truck(const Vector &direction) {  // Camera.cpp:33-37
  const Vector move = direction * rotationMatrix;
  position += move;
  return *this;
}

Camera &Camera::pan(const float degrees) {  // Camera.cpp:39-48
  const float r = degreesToRadians(degrees);
  Matrix3 rot;
  rot.m[0][0] = cosf(r); rot.m[0][1] = 0.0f; rot.m[0][2] = -sinf(r);
  rot.m[1][0] = 0.0f;    rot.m[1][1] = 1.0f; rot.m[1][2] = 0.0f;
  rot.m[2][0] = sinf(r); rot.m[2][1] = 0.0f; rot.m[2][2] = cosf(r);
  rotationMatrix = rotationMatrix * rot;
  return *this;
}

Camera &Camera::roll(const float degrees) {  // Camera.cpp:50-59
  const float r = degreesToRadians(degrees);
  Matrix3 rot;
  rot.m[0][0] = cosf(r); rot.m[0][1] = -sinf(r); rot.m[0][2] = 0.0f;
  rot.m[1][0] = sinf(r); rot.m[1][1] = cosf(r);  rot.m[1][2] = 0.0f;
  rot.m[2][0] = 0.0f;    rot.m[2][1] = 0.0f;     rot.m[2][2] = 1.0f;
  rotationMatrix = rotationMatrix * rot;
  return *this;
}

Camera &Camera::tilt(const float degrees) {  // Camera.cpp:61-70
  const float r = degreesToRadians(degrees);
  Matrix3 rot;
  rot.m[0][0] = 1.0f; rot.m[0][1] = 0.0f;    rot.m[0][2] = 0.0f;
  rot.m[1][0] = 0.0f; rot.m[1][1] = cosf(r); rot.m[1][2] = -sinf(r);
  rot.m[2][0] = 0.0f; rot.m[2][1] = sinf(r); rot.m[2][2] = cosf(r);
  rotationMatrix = rotationMatrix * rot;
  return *this;
}

// reference: Scene.cpp:5-30 + Triangle.cpp:13-16,22-27.  Face normal = normalised (v1-v0) x (v2-v0);
// vertex normal = normalised unweighted sum of the face normals of the triangles that use the vertex.
Mesh::Mesh(unsigned int material, const std::vector<Vertex> &verts, const std::vector<unsigned int> &indexes)
    : material(material), vertices(verts) {
  triangles.reserve(indexes.size() / 3);
  for (size_t i = 0; i + 2 < indexes.size(); i += 3) {
    Triangle t;
    t.indexes[0] = indexes[i];
    t.indexes[1] = indexes[i + 1];
    t.indexes[2] = indexes[i + 2];
    const Vector e1 = vertices[t.indexes[1]].position - vertices[t.indexes[0]].position;
    const Vector e2 = vertices[t.indexes[2]].position - vertices[t.indexes[0]].position;
    t.normal = e1.cross(e2);
    t.normal.normalize();
    triangles.push_back(t);
    vertices[t.indexes[0]].normal += t.normal;
    vertices[t.indexes[1]].normal += t.normal;
    vertices[t.indexes[2]].normal += t.normal;
  }
  for (auto &v : vertices) v.normal.normalize();
}

}  // namespace crt
