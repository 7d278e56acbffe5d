// Member declarations of the reference's types AS THE BINDING READS THEM -- names and types follow the cited reference
// headers -- so that tests/cpp/rgk_binding.inc (the code INTEGRATION.md shows) compiles here exactly as it would inside
// the reference tree.  Declarations only: no reference function bodies, nothing of the renderer.
//   glm::vec2/vec3            GLM (packed floats x,y,z)
//   Color, Radiance           src/radiance.hpp:6-60      {r,g,b}
//   ReadableTexture family    src/texture.hpp:10-80      (+ `friend struct RgkBinding` on FileTexture: data/xsize/ysize are private)
//   EXRTexture                src/texture.hpp:83-118     (+ friend: data/count are private)
//   Material, BxDF subclasses src/bxdf/bxdf.hpp:19-159
//   LTCdef, mat33             src/LTC/ltc.hpp:4-18
//   Light, Triangle           src/primitives.hpp:26-43,65-95
//   Scene                     src/scene.hpp:76-172       (+ friend: skybox_* are private)
//   Camera                    src/camera.hpp:27-41
//   RenderTask                src/tracer.hpp:14-24
//   Config                    src/config.hpp:26-42
#pragma once
#include <cmath>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace glm {
struct vec2 { float x, y; vec2(float x_ = 0, float y_ = 0) : x(x_), y(y_) {} };
struct vec3 { float x, y, z; vec3(float x_ = 0, float y_ = 0, float z_ = 0) : x(x_), y(y_), z(z_) {} };
} // namespace glm

struct RgkBinding;

struct Color { float r, g, b; Color(float r_ = 0, float g_ = 0, float b_ = 0) : r(r_), g(g_), b(b_) {} };
struct Radiance { float r, g, b; Radiance(float r_ = 0, float g_ = 0, float b_ = 0) : r(r_), g(g_), b(b_) {} };

class ReadableTexture {
public:
    virtual ~ReadableTexture() {}
    virtual Color GetPixel(int x, int y) const = 0;
    virtual bool Empty() const = 0;
};
class FileTexture : public ReadableTexture {
public:
    FileTexture(int xs, int ys) : data((size_t)xs * ys), xsize(xs), ysize(ys) {}
    void SetPixel(int x, int y, Color c) { data[(size_t)y * xsize + x] = c; }
    Color GetPixel(int x, int y) const override { return data[(size_t)y * xsize + x]; }
    bool Empty() const override { return false; }
private:
    std::vector<Color> data;
    unsigned int xsize, ysize;
    friend struct RgkBinding; // the one edit the binding needs in src/texture.hpp
};
class SolidTexture : public ReadableTexture {
public:
    SolidTexture(Color c) : color(c) {}
    Color GetPixel(int, int) const override { return color; }
    bool Empty() const override { return false; }
private:
    Color color;
};
class EmptyTexture : public SolidTexture {
public:
    EmptyTexture() : SolidTexture(Color(0, 0, 0)) {}
    bool Empty() const override { return true; }
};

class EXRTexture {
public:
    EXRTexture(int xs = 0, int ys = 0) : xsize(xs), ysize(ys), data((size_t)xs * ys), count((size_t)xs * ys, 0u) {}
private:
    unsigned int xsize, ysize;
    std::vector<Radiance> data;
    std::vector<unsigned int> count;
    friend struct RgkBinding; // src/texture.hpp:112-115
    friend int main(int, char**);
};

struct mat33 { double m[9]; };
struct LTCdef { const int size; const mat33* tabM; const float* tabAmplitude; };
class LTC { public: static const LTCdef Beckmann; static const LTCdef GGX; };

class BxDF { public: virtual ~BxDF() {} };
class Material {
public:
    std::string name;
    Radiance emission;
    std::shared_ptr<ReadableTexture> bumpmap;
    std::unique_ptr<BxDF> bxdf;
    bool is_thinglass = false;
    bool no_russian = false;
};
class BxDFDiffuse : public BxDF { public: std::shared_ptr<ReadableTexture> diffuse = std::make_shared<EmptyTexture>(); };
class BxDFTransparent : public BxDF {};
class BxDFMirror : public BxDF { public: std::shared_ptr<ReadableTexture> color = std::make_shared<EmptyTexture>(); };
class BxDFDielectric : public BxDF { public: float ior = 1.0; std::shared_ptr<ReadableTexture> color = std::make_shared<EmptyTexture>(); };
class BxDFMix : public BxDF { public: std::shared_ptr<const Material> m1, m2; float amt1; };
class BxDFLTCBase : public BxDF { public: float roughness; std::shared_ptr<ReadableTexture> color = std::make_shared<EmptyTexture>(); };
class BxDFLTCDiffuseBase : public BxDFLTCBase { public: std::shared_ptr<ReadableTexture> diffuse = std::make_shared<EmptyTexture>(); };
template <const LTCdef& ltc> class BxDFLTC : public BxDFLTCBase {};
template <const LTCdef& ltc> class BxDFLTCDiffuse : public BxDFLTCDiffuseBase {};

struct Light {
    enum Type { FULL_SPHERE, HEMISPHERE };
    Light(Type t) : type(t) {}
    Type type;
    glm::vec3 pos;
    Radiance color;
    float intensity;
    float size;
    glm::vec3 normal;
};

class Scene;
class Triangle {
public:
    const Scene* parent_scene;
    unsigned int va, vb, vc;
    Material* mat;
    Triangle(const Scene* parent, unsigned int a, unsigned int b, unsigned int c, Material* m) : parent_scene(parent), va(a), vb(b), vc(c), mat(m) {}
    Triangle() : parent_scene(nullptr) {}
};

class Scene {
public:
    glm::vec3* vertices = nullptr;   unsigned int n_vertices = 0;
    Triangle* triangles = nullptr;   unsigned int n_triangles = 0;
    glm::vec3* normals = nullptr;    unsigned int n_normals = 0;
    glm::vec3* tangents = nullptr;   unsigned int n_tangents = 0;
    glm::vec2* texcoords = nullptr;  unsigned int n_texcoords = 0;
    std::vector<Light> pointlights;
    struct ArealLight {
        std::vector<std::pair<float, unsigned int>> triangles_with_areas;
        mutable float total_area = 0.0f;
        Radiance emission;
        float power = 0.0f;
    };
    std::vector<std::pair<float, ArealLight>> areal_lights;
    float epsilon = 0.0001f;
    void SetSkyboxColor(Color c, float intensity) { skybox_mode = SimpleRadiance; skybox_color = c; skybox_intensity = intensity; }
    void SetSkyboxTexture(std::shared_ptr<ReadableTexture> t, float intensity, float rotate) { // SetSkyboxEnvmap minus the file load
        skybox_mode = Envmap; skybox_texture = t; skybox_intensity = intensity; skybox_rotate = rotate;
    }
private:
    enum SkyboxMode : int { SimpleRadiance, Envmap };
    SkyboxMode skybox_mode = SimpleRadiance;
    Color skybox_color;
    std::shared_ptr<ReadableTexture> skybox_texture;
    float skybox_intensity = 1.0f;
    float skybox_rotate = 0.0f;
    friend struct RgkBinding; // src/scene.hpp:160-172
};

class Camera {
public:
    glm::vec3 origin, lookat, direction, cameraup, cameraleft, viewscreen, viewscreen_x, viewscreen_y;
    float lens_size;
    int xsize, ysize;
};

struct RenderTask {
    RenderTask(unsigned int xr, unsigned int yr, unsigned int x1, unsigned int x2, unsigned int y1, unsigned int y2)
        : xres(xr), yres(yr), xrange_start(x1), xrange_end(x2), yrange_start(y1), yrange_end(y2) {
        midpoint = glm::vec2((xrange_start + xrange_end) / 2.0f, (yrange_start + yrange_end) / 2.0f);
    }
    unsigned int xres, yres;
    unsigned int xrange_start, xrange_end;
    unsigned int yrange_start, yrange_end;
    glm::vec2 midpoint;
};

class Config {
public:
    unsigned int recursion_level = 40;
    unsigned int xres, yres;
    unsigned int multisample = 1;
    float bumpmap_scale = 10.0f;
    float clamp = 100000.0f;
    float russian = -1.0f;
    float output_scale = -1.0f;
    unsigned int render_rounds = 1;
    bool force_fresnell = false;
    unsigned int reverse = 0;
};
