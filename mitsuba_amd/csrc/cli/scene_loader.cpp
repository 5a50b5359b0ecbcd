#include "scene_loader.h"
#include "xml_lite.h"

#include <sys/stat.h>
#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <map>
#include <sstream>
#include <tuple>

namespace mtsamd_host {

namespace {

typedef xml_lite::Element Element;

struct Fail : std::runtime_error { using std::runtime_error::runtime_error; };

std::string lower(std::string s) { for (char &c : s) c = (char) std::tolower((unsigned char) c); return s; }

std::string dirOf(const std::string &path) {
    size_t p = path.find_last_of('/');
    return p == std::string::npos ? std::string(".") : path.substr(0, p);
}

std::string joinPath(const std::string &dir, const std::string &file) {
    if (!file.empty() && file[0] == '/') return file;
    return dir + "/" + file;
}

bool readFile(const std::string &path, std::vector<uint8_t> &out) {
    struct stat info;
    if (stat(path.c_str(), &info) != 0 || !S_ISREG(info.st_mode)) return false;     // a directory opens, and reports a size of 2^63 - 1
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    f.seekg(0, std::ios::end);
    std::streamoff n = f.tellg();
    f.seekg(0);
    if (n < 0) return false;
    out.resize((size_t) n);
    if (n) f.read((char *) out.data(), n);
    return (bool) f;
}

std::vector<double> numbers(const std::string &s) {
    std::vector<double> out;
    std::string t = s;
    for (char &c : t) if (c == ',') c = ' ';
    std::istringstream is(t);
    double v;
    while (is >> v) out.push_back(v);
    return out;
}

// ---- 4x4 matrices in double, row-major ------------------------------------------------------------------------------
struct Mat4 {
    double m[16];
    static Mat4 identity() { Mat4 r; for (int i = 0; i < 16; ++i) r.m[i] = (i % 5 == 0) ? 1.0 : 0.0; return r; }
    Mat4 operator*(const Mat4 &b) const {
        Mat4 r;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                double s = 0;
                for (int k = 0; k < 4; ++k) s += m[4 * i + k] * b.m[4 * k + j];
                r.m[4 * i + j] = s;
            }
        return r;
    }
    bool isIdentity() const { Mat4 i = identity(); return memcmp(m, i.m, sizeof(m)) == 0; }
};

bool invert(const Mat4 &a, Mat4 &out) {
    double w[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { w[i][j] = a.m[4 * i + j]; w[i][j + 4] = i == j ? 1.0 : 0.0; }
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        for (int r = c + 1; r < 4; ++r) if (std::fabs(w[r][c]) > std::fabs(w[piv][c])) piv = r;
        if (w[piv][c] == 0) return false;
        if (piv != c) for (int j = 0; j < 8; ++j) std::swap(w[c][j], w[piv][j]);
        const double inv = 1.0 / w[c][c];
        for (int j = 0; j < 8; ++j) w[c][j] *= inv;
        for (int r = 0; r < 4; ++r) {
            if (r == c) continue;
            const double f = w[r][c];
            if (f != 0) for (int j = 0; j < 8; ++j) w[r][j] -= f * w[c][j];
        }
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) out.m[4 * i + j] = w[i][j + 4];
    return true;
}

// Transform::lookAt (src/libcore/transform.cpp:191-214), single precision like the reference
Mat4 lookAt(const std::vector<double> &o, const std::vector<double> &t, const std::vector<double> &up) {
    float d[3] = { (float) t[0] - (float) o[0], (float) t[1] - (float) o[1], (float) t[2] - (float) o[2] };
    float len = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (len == 0) throw Fail("lookAt(): 'origin' and 'target' coincide!");
    for (float &c : d) c /= len;
    const float u[3] = { (float) up[0], (float) up[1], (float) up[2] };
    float left[3] = { u[1] * d[2] - u[2] * d[1], u[2] * d[0] - u[0] * d[2], u[0] * d[1] - u[1] * d[0] };
    len = std::sqrt(left[0] * left[0] + left[1] * left[1] + left[2] * left[2]);
    if (len == 0) throw Fail("lookAt(): the forward and upward direction must be linearly independent!");
    for (float &c : left) c /= len;
    const float nu[3] = { d[1] * left[2] - d[2] * left[1], d[2] * left[0] - d[0] * left[2], d[0] * left[1] - d[1] * left[0] };
    Mat4 r = Mat4::identity();
    for (int i = 0; i < 3; ++i) { r.m[4 * i + 0] = left[i]; r.m[4 * i + 1] = nu[i]; r.m[4 * i + 2] = d[i]; r.m[4 * i + 3] = (float) o[i]; }
    return r;
}

// <transform> children are applied in order, each one multiplied from the left (scenehandler.cpp:574-640)
Mat4 parseTransform(const Element &node) {
    Mat4 m = Mat4::identity();
    for (const auto &cp : node.children) {
        const Element &c = *cp;
        Mat4 t = Mat4::identity();
        auto xyz = [&](double def, double out[3]) {
            const char *k[3] = { "x", "y", "z" };
            for (int i = 0; i < 3; ++i) out[i] = c.has(k[i]) ? std::stod(c.get(k[i])) : def;
        };
        if (c.tag == "matrix") {
            std::vector<double> f = numbers(c.get("value"));
            if (f.size() != 16) throw Fail("<matrix> needs 16 values");
            for (int i = 0; i < 16; ++i) t.m[i] = f[i];
        } else if (c.tag == "lookat") {
            std::vector<double> o = numbers(c.get("origin")), tg = numbers(c.get("target")), up = numbers(c.get("up", "0,0,1"));
            if (o.size() != 3 || tg.size() != 3 || up.size() != 3) throw Fail("<lookat> needs origin, target (and up) with three values each");
            t = lookAt(o, tg, up);
        } else if (c.tag == "translate") {
            double v[3]; xyz(0.0, v);
            t.m[3] = v[0]; t.m[7] = v[1]; t.m[11] = v[2];
        } else if (c.tag == "scale") {
            double v[3];
            if (c.has("value")) v[0] = v[1] = v[2] = std::stod(c.get("value")); else xyz(1.0, v);
            t.m[0] = v[0]; t.m[5] = v[1]; t.m[10] = v[2];
        } else if (c.tag == "rotate") {     // Transform::rotate (transform.cpp:65-98)
            double ax[3]; xyz(0.0, ax);
            const double n = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
            if (n == 0 || !c.has("angle")) throw Fail("<rotate> needs an axis and an angle");
            const double x = ax[0] / n, y = ax[1] / n, z = ax[2] / n;
            const double a = std::stod(c.get("angle")) * 3.14159265358979323846 / 180.0, cs = std::cos(a), sn = std::sin(a);
            t.m[0] = cs + x * x * (1 - cs); t.m[1] = x * y * (1 - cs) - z * sn; t.m[2] = x * z * (1 - cs) + y * sn;
            t.m[4] = y * x * (1 - cs) + z * sn; t.m[5] = cs + y * y * (1 - cs); t.m[6] = y * z * (1 - cs) - x * sn;
            t.m[8] = z * x * (1 - cs) - y * sn; t.m[9] = z * y * (1 - cs) + x * sn; t.m[10] = cs + z * z * (1 - cs);
        } else {
            throw Fail("unsupported transform element <" + c.tag + ">");
        }
        m = t * m;
    }
    return m;
}

// ---- properties ---------------------------------------------------------------------------------------------------------
struct Value {
    enum Kind { Float, Int, Bool, String, Rgb, Transform, Vec3 } kind = Float;     // Vec3 (<point>, <vector>) uses `rgb`
    double f = 0; long i = 0; bool b = false; std::string s; float rgb[3] = { 0, 0, 0 }; Mat4 t;
};
typedef std::map<std::string, Value> Props;

float srgbToLinear(float c) { return c <= 0.04045f ? c / 12.92f : std::pow((c + 0.055f) / 1.055f, 2.4f); }

// Directory against which <spectrum filename=...> is resolved: the scene file's (set by loadSceneXml)
thread_local std::string g_spectrumDir;

// scenehandler.cpp:503-624 in the three-sample build (where Spectrum::getD65() is 1, so that both intents of a single
// value give the same constant)
void parseSpectrum(const Element &n, float out[3]) {
    const std::string v = n.get("value");
    if (n.tag == "rgb") {
        std::vector<double> f = numbers(v);
        if (f.size() == 1) f = { f[0], f[0], f[0] };
        if (f.size() < 3) throw Fail("<rgb> needs one or three values");
        for (int i = 0; i < 3; ++i) out[i] = (float) f[i];
    } else if (n.tag == "srgb") {
        float f[3];
        if (!v.empty() && v[0] == '#') {
            if (v.size() != 7) throw Fail("<srgb> hex value must look like #rrggbb");
            for (int i = 0; i < 3; ++i) f[i] = (float) std::stoi(v.substr(1 + 2 * i, 2), nullptr, 16) / 255.0f;
        } else {
            std::vector<double> g = numbers(v);
            if (g.size() == 1) g = { g[0], g[0], g[0] };
            if (g.size() < 3) throw Fail("<srgb> needs one or three values");
            for (int i = 0; i < 3; ++i) f[i] = (float) g[i];
        }
        for (int i = 0; i < 3; ++i) out[i] = srgbToLinear(f[i]);
    } else if (n.tag == "blackbody") {
        std::string t = n.get("temperature");
        while (!t.empty() && isspace((unsigned char) t.back())) t.pop_back();
        if (!t.empty() && toupper((unsigned char) t.back()) == 'K') t.pop_back();
        if (t.empty()) throw Fail("<blackbody> needs a temperature");
        const float scale = n.has("scale") ? (float) std::stod(n.get("scale")) : 1.0f;
        if (mtsamd_blackbody_to_rgb((float) std::stod(t), scale, out) != MTSAMD_OK) throw Fail("<blackbody>: the temperature must be positive");
    } else if (n.tag == "spectrum") {
        if (n.has("value") == n.has("filename")) throw Fail("<spectrum>: please provide one of 'value' or 'filename'");
        if (n.has("filename")) {
            if (n.has("intent")) throw Fail("<spectrum>: 'intent' and 'filename' cannot be specified at the same time!");
            std::string file = n.get("filename");
            if (!file.empty() && file[0] != '/' && !g_spectrumDir.empty()) file = g_spectrumDir + "/" + file;
            if (mtsamd_spectrum_file_to_rgb(file.c_str(), out) != MTSAMD_OK) throw Fail("InterpolatedSpectrum: could not read \"" + file + "\"");
            return;
        }
        if (v.find(':') == std::string::npos) {
            const std::vector<double> f = numbers(v);
            if (f.size() == 1) {
                if (n.has("intent")) {
                    const std::string intent = lower(n.get("intent"));
                    if (intent != "reflectance" && intent != "illuminant")
                        throw Fail("Invalid intent \"" + intent + "\", must be \"reflectance\" or \"illuminant\"");
                }
                out[0] = out[1] = out[2] = (float) f[0];
                return;
            }
            if (n.has("intent")) throw Fail("<spectrum>: 'intent' can only be specified when given a single-valued argument.");
            if (f.size() != 3) throw Fail("Invalid spectrum value specified (length does not match the current spectral discretization!)");
            for (int i = 0; i < 3; ++i) out[i] = (float) f[i];
            return;
        }
        if (n.has("intent")) throw Fail("<spectrum>: 'intent' can only be specified when given a single-valued argument.");
        std::vector<float> lambda, value;       // wavelength:value pairs separated by commas and / or blanks
        std::string token;
        auto flush = [&]() {
            if (token.empty()) return;
            const size_t colon = token.find(':');
            if (colon == std::string::npos || token.find(':', colon + 1) != std::string::npos || colon == 0 || colon + 1 == token.size())
                throw Fail("Invalid spectrum->value mapping specified");
            lambda.push_back((float) std::stod(token.substr(0, colon)));
            value.push_back((float) std::stod(token.substr(colon + 1)));
            token.clear();
        };
        for (char c : v) { if (c == ',' || c == ' ' || c == '\t' || c == '\n') flush(); else token.push_back(c); }
        flush();
        if (lambda.size() < 2) throw Fail("InterpolatedSpectrum::zeroExtend() -- at least 2 entries are needed!");
        if (mtsamd_spectrum_value_to_rgb(lambda.data(), value.data(), (uint32_t) lambda.size(), out) != MTSAMD_OK)
            throw Fail("InterpolatedSpectrum: spectral power distribution values must be provided in order of increasing wavelength!");
    } else {
        throw Fail("unsupported spectrum tag <" + n.tag + ">");
    }
}

Props parseProps(const Element &node) {
    Props out;
    for (const auto &cp : node.children) {
        const Element &c = *cp;
        Value v;
        const std::string name = c.get("name");
        if (c.tag == "float") { v.kind = Value::Float; v.f = std::stod(c.get("value")); }
        else if (c.tag == "integer") { v.kind = Value::Int; v.i = std::stol(c.get("value")); v.f = (double) v.i; }
        else if (c.tag == "boolean") { v.kind = Value::Bool; v.b = lower(c.get("value")) == "true"; }
        else if (c.tag == "string") { v.kind = Value::String; v.s = c.get("value"); }
        else if (c.tag == "rgb" || c.tag == "srgb" || c.tag == "spectrum" || c.tag == "blackbody") { v.kind = Value::Rgb; parseSpectrum(c, v.rgb); }
        else if (c.tag == "transform") { v.kind = Value::Transform; v.t = parseTransform(c); }
        else if (c.tag == "point" || c.tag == "vector") {   // scenehandler.cpp:443-459
            v.kind = Value::Vec3;
            const char *axes[3] = { "x", "y", "z" };
            for (int i = 0; i < 3; ++i) v.rgb[i] = c.has(axes[i]) ? (float) std::stod(c.get(axes[i])) : 0.0f;
        }
        else continue;
        out[name] = v;
    }
    return out;
}

double getF(const Props &p, const char *k, double def) { auto it = p.find(k); return it == p.end() ? def : it->second.f; }
bool getB(const Props &p, const char *k, bool def) { auto it = p.find(k); return it == p.end() ? def : it->second.b; }
std::string getS(const Props &p, const char *k, const std::string &def) { auto it = p.find(k); return it == p.end() ? def : it->second.s; }
void getRgb(const Props &p, const char *k, float d0, float d1, float d2, float out[3]) {
    auto it = p.find(k);
    if (it == p.end()) { out[0] = d0; out[1] = d1; out[2] = d2; }
    else if (it->second.kind == Value::Rgb) memcpy(out, it->second.rgb, 12);
    else out[0] = out[1] = out[2] = (float) it->second.f;
}

// lookupIOR (src/bsdfs/ior.h:39-111)
float lookupIor(const Props &p, const char *key, const char *defName) {
    static const std::pair<const char *, float> table[] = {
        { "vacuum", 1.0f }, { "helium", 1.000036f }, { "hydrogen", 1.000132f }, { "air", 1.000277f }, { "carbon dioxide", 1.00045f },
        { "water", 1.3330f }, { "acetone", 1.36f }, { "ethanol", 1.361f }, { "carbon tetrachloride", 1.461f }, { "glycerol", 1.4729f },
        { "benzene", 1.501f }, { "silicone oil", 1.52045f }, { "bromine", 1.661f }, { "water ice", 1.31f }, { "fused quartz", 1.458f },
        { "pyrex", 1.470f }, { "acrylic glass", 1.49f }, { "polypropylene", 1.49f }, { "bk7", 1.5046f }, { "sodium chloride", 1.544f },
        { "amber", 1.55f }, { "pet", 1.5750f }, { "diamond", 2.419f } };
    auto it = p.find(key);
    if (it != p.end() && it->second.kind != Value::String) return (float) it->second.f;
    const std::string name = lower(it == p.end() ? std::string(defName) : it->second.s);
    for (const auto &e : table) if (name == e.first) return e.second;
    throw Fail("Unable to find an IOR value for \"" + name + "\"!");
}

// eta and k of the conductor models before the division by extEta (src/bsdfs/roughconductor.cpp:174-190,
// src/bsdfs/conductor.cpp:153-171): explicit values win, "none" stands for (0, 1), anything else names the measured
// spectra data/ior/<material>.{eta,k}.spd (default copper)
void conductorConstants(const Props &p, const LoadOptions &opts, mtsamd_material &m) {
    const std::string material = getS(p, "material", "Cu");
    const bool none = lower(material) == "none";
    struct { const char *key; float noneValue; float *out; } parts[2] = { { "eta", 0.0f, m.eta }, { "k", 1.0f, m.k } };
    for (auto &part : parts) {
        if (p.count(part.key) || none) { getRgb(p, part.key, part.noneValue, part.noneValue, part.noneValue, part.out); continue; }
        if (opts.dataDir.empty())
            throw Fail("conductor material preset '" + material + "' needs the measured spectra: pass the directory that holds ior/"
                       + material + "." + part.key + ".spd");
        const std::string path = opts.dataDir + "/ior/" + material + "." + part.key + ".spd";
        if (mtsamd_spd_file_to_rgb(path.c_str(), part.out) != MTSAMD_OK)
            throw Fail("InterpolatedSpectrum: could not read \"" + path + "\"");
    }
}

// ---- textures ----------------------------------------------------------------------------------------------------------------
// A <texture> element as a record of the scene's texture list; returns its 1-based index. `scale` textures
// (src/textures/scale.cpp:56-66) fold into the `scale` factor of a copy of the texture they wrap.
uint32_t makeTexture(const Element &node, LoadedScene &out) {
    const std::string type = node.get("type");
    const Props p = parseProps(node);
    mtsamd_texture t;
    memset(&t, 0, sizeof(t));
    t.scale = 1.0f;
    std::vector<float> pixels;
    auto uvTransform = [&]() {          // Texture2D (src/librender/texture.cpp:81-95)
        t.u_offset = (float) getF(p, "uoffset", 0.0); t.v_offset = (float) getF(p, "voffset", 0.0);
        t.u_scale = (float) getF(p, "uscale", 1.0); t.v_scale = (float) getF(p, "vscale", 1.0);
    };
    if (type == "scale") {
        const Element *nested = nullptr;
        for (const auto &c : node.children)
            if (c->tag == "texture" || c->tag == "ref") { if (nested) throw Fail("scale texture: exactly one nested texture is expected"); nested = c.get(); }
        if (!nested) throw Fail("scale texture: a nested texture is required");
        uint32_t inner;
        if (nested->tag == "ref") {
            if (!out.namedTextures.count(nested->get("id"))) throw Fail("reference to unknown texture '" + nested->get("id") + "'");
            inner = out.namedTextures.at(nested->get("id"));
        } else inner = makeTexture(*nested, out);
        float factor[3];
        getRgb(p, "scale", 1.0f, 1.0f, 1.0f, factor);
        if (factor[0] != factor[1] || factor[0] != factor[2]) throw Fail("scale texture: only a scalar (or grey) scale factor is supported");
        t = out.textures[inner - 1];
        pixels = out.texturePixels[inner - 1];
        t.scale *= factor[0];
    } else if (type == "checkerboard") {        // src/textures/checkerboard.cpp:48-51
        t.type = MTSAMD_TEX_CHECKERBOARD;
        getRgb(p, "color0", 0.4f, 0.4f, 0.4f, t.color0); getRgb(p, "color1", 0.2f, 0.2f, 0.2f, t.color1);
        uvTransform();
    } else if (type == "gridtexture") {         // src/textures/gridtexture.cpp:55-59
        t.type = MTSAMD_TEX_GRID;
        getRgb(p, "color0", 0.2f, 0.2f, 0.2f, t.color0); getRgb(p, "color1", 0.4f, 0.4f, 0.4f, t.color1);
        t.line_width = (float) getF(p, "lineWidth", 0.01);
        uvTransform();
    } else if (type == "bitmap") {              // src/textures/bitmap.cpp:179-235
        t.type = MTSAMD_TEX_BITMAP;
        uvTransform();
        if (!p.count("filename")) throw Fail("bitmap texture: the 'filename' parameter is missing");
        if (!getS(p, "channel", "").empty()) throw Fail("bitmap texture: the 'channel' parameter is not supported");
        std::string file = getS(p, "filename", "");
        if (!file.empty() && file[0] != '/') file = out.sceneDir + "/" + file;
        const std::string filter = lower(getS(p, "filterType", "ewa"));
        if (filter == "ewa") t.filter_type = MTSAMD_MIP_EWA; else if (filter == "bilinear") t.filter_type = MTSAMD_MIP_BILINEAR;
        else if (filter == "trilinear") t.filter_type = MTSAMD_MIP_TRILINEAR; else if (filter == "nearest") t.filter_type = MTSAMD_MIP_NEAREST;
        else throw Fail("Unknown filter type '" + filter + "' -- must be 'ewa', 'trilinear', or 'nearest'!");
        auto wrap = [&](const std::string &mode) {      // bitmap.cpp:324-339
            if (mode == "repeat") return (int) MTSAMD_WRAP_REPEAT;
            if (mode == "clamp") return (int) MTSAMD_WRAP_CLAMP;
            if (mode == "mirror") return (int) MTSAMD_WRAP_MIRROR;
            if (mode == "zero" || mode == "black") return (int) MTSAMD_WRAP_ZERO;
            if (mode == "one" || mode == "white") return (int) MTSAMD_WRAP_ONE;
            throw Fail("Unknown wrap mode '" + mode + "' -- must be 'repeat', 'clamp', 'black', or 'white'!");
        };
        const std::string both = getS(p, "wrapMode", "repeat");
        t.wrap_u = wrap(getS(p, "wrapModeU", both)); t.wrap_v = wrap(getS(p, "wrapModeV", both));
        t.max_anisotropy = (float) getF(p, "maxAnisotropy", 20.0);
        mtsamd_image img;
        if (mtsamd_image_load(file.c_str(), (float) getF(p, "gamma", 0.0), &img) != MTSAMD_OK) throw Fail(mtsamd_last_error());
        pixels.assign(img.pixels, img.pixels + (size_t) img.width * img.height * img.channels);
        t.width = img.width; t.height = img.height; t.channels = img.channels;
        mtsamd_image_free(&img);
    } else {
        throw Fail("texture type '" + type + "' is not supported (checkerboard, gridtexture, bitmap, scale)");
    }
    out.textures.push_back(t);
    out.texturePixels.push_back(std::move(pixels));
    return (uint32_t) out.textures.size();
}

// The textures a <bsdf> element names for its colour parameters: nested <texture name="..."> elements and
// <ref name="..." id="..."/> to textures declared at the top level
void attachTextures(const Element &node, LoadedScene &out, mtsamd_material &m) {
    for (const auto &c : node.children) {
        uint32_t index = 0;
        if (c->tag == "texture") index = makeTexture(*c, out);
        else if (c->tag == "ref" && out.namedTextures.count(c->get("id"))) index = out.namedTextures.at(c->get("id"));
        else continue;
        const std::string name = c->get("name");
        const bool diffuseName = name == "diffuseReflectance" || (name == "reflectance" && m.type == MTSAMD_BSDF_DIFFUSE);
        const bool usesDiffuse = m.type == MTSAMD_BSDF_DIFFUSE || m.type == MTSAMD_BSDF_ROUGHPLASTIC || m.type == MTSAMD_BSDF_PLASTIC || m.type == MTSAMD_BSDF_WARD;
        const bool usesTransmittance = m.type == MTSAMD_BSDF_DIELECTRIC || m.type == MTSAMD_BSDF_ROUGHDIELECTRIC;
        const bool usesRoughnessTexture = m.type == MTSAMD_BSDF_ROUGHCONDUCTOR || m.type == MTSAMD_BSDF_ROUGHDIELECTRIC;
        // roughplastic.cpp:218: one isotropic roughness; its transmittance table was kept two-dimensional by makeBsdf
        const bool plasticRoughness = m.type == MTSAMD_BSDF_ROUGHPLASTIC && name == "alpha";
        if (diffuseName && usesDiffuse) m.diffuse_texture = (int32_t) index;
        else if (name == "specularReflectance" && m.type != MTSAMD_BSDF_DIFFUSE) m.specular_texture = (int32_t) index;
        else if (name == "specularTransmittance" && usesTransmittance) m.transmittance_texture = (int32_t) index;
        // roughconductor.cpp:207-221, roughdielectric.cpp:224-243: alpha sets both directions
        else if (name == "alpha" && (usesRoughnessTexture || plasticRoughness)) m.alpha_u_texture = m.alpha_v_texture = (int32_t) index;
        else if (name == "alphaU" && usesRoughnessTexture) m.alpha_u_texture = (int32_t) index;
        else if (name == "alphaV" && usesRoughnessTexture) m.alpha_v_texture = (int32_t) index;
        else throw Fail("a texture for the parameter '" + name + "' of <bsdf type=\"" + node.get("type") + "\"> is not supported");
    }
}

// ---- BSDFs -------------------------------------------------------------------------------------------------------------------
uint32_t addMaterial(LoadedScene &out, const mtsamd_material &m);

// A copy of an already added record for use as a nested BRDF. Rough-transmittance tables are matched to the
// roughplastic records by order of appearance, so the copy brings a copy of its table.
mtsamd_material copyMaterial(LoadedScene &out, uint32_t index) {
    const mtsamd_material m = out.materials[index];
    if (m.type == MTSAMD_BSDF_ROUGHPLASTIC) {
        size_t ordinal = 0;
        for (uint32_t i = 0; i < index; ++i) ordinal += out.materials[i].type == MTSAMD_BSDF_ROUGHPLASTIC;
        const std::vector<float> table = out.rtransTables[ordinal];
        out.rtransTables.push_back(table);
    }
    return m;
}

mtsamd_material makeBsdf(const Element &node, const LoadOptions &opts, LoadedScene &out, const std::map<std::string, uint32_t> *named = nullptr) {
    const std::string type = node.get("type");
    if (type == "twosided") {                   // src/bsdfs/twosided.cpp:84-106,185-196
        std::vector<const Element *> nested;
        for (const auto &c : node.children)
            if (c->tag == "bsdf" || c->tag == "ref") nested.push_back(c.get());
        if (nested.empty()) throw Fail("A nested one-sided material is required!");
        if (nested.size() > 2) throw Fail("No more than two nested BRDFs can be added!");
        auto make = [&](const Element &c) {
            if (c.tag == "ref") {
                if (!named || !named->count(c.get("id"))) throw Fail("reference to unknown bsdf '" + c.get("id") + "'");
                return copyMaterial(out, named->at(c.get("id")));
            }
            if (c.get("type") == "twosided") throw Fail("a <bsdf type=\"twosided\"> nested in another one is not supported");
            return makeBsdf(c, opts, out, named);
        };
        // the record of the back side is appended first, so that records and their tables stay in the same order
        int32_t back = -1;
        if (nested.size() == 2) {
            const mtsamd_material b = make(*nested[1]);
            if (b.type == MTSAMD_BSDF_DIELECTRIC || b.type == MTSAMD_BSDF_ROUGHDIELECTRIC) throw Fail("Only materials without a transmission component can be nested!");
            back = (int32_t) addMaterial(out, b);
        }
        mtsamd_material m = make(*nested[0]);
        if (m.type == MTSAMD_BSDF_DIELECTRIC || m.type == MTSAMD_BSDF_ROUGHDIELECTRIC) throw Fail("Only materials without a transmission component can be nested!");
        m.two_sided = 1;
        m.back_material = back;
        return m;
    }
    if (type == "bumpmap" || type == "normalmap") {     // src/bsdfs/bumpmap.cpp:116-133, src/bsdfs/normalmap.cpp:88-105
        const Element *nestedBsdf = nullptr, *texture = nullptr;
        for (const auto &c : node.children) {
            const bool isTextureRef = c->tag == "ref" && out.namedTextures.count(c->get("id"));
            if (c->tag == "texture" || isTextureRef) {
                if (texture) throw Fail(type == "bumpmap" ? "Only a single displacement texture can be specified!" : "Only a single normal texture can be specified!");
                texture = c.get();
            } else if (c->tag == "bsdf" || c->tag == "ref") {
                if (nestedBsdf) throw Fail("Only a single nested BSDF can be added!");
                nestedBsdf = c.get();
            }
        }
        if (!nestedBsdf) throw Fail("A child BSDF instance is required");
        if (!texture) throw Fail(type == "bumpmap" ? "A displacement texture must be specified" : "A normal map texture must be specified");
        if (texture->tag == "texture" && texture->get("type") == "bitmap") {
            bool hasGamma = false;
            for (const auto &c : texture->children) hasGamma |= c->get("name") == "gamma";
            if (!hasGamma) throw Fail("When using a bitmap texture as a bump map, please explicitly specify the 'gamma' parameter of the bitmap plugin. "
                                      "In most cases the following is the correct choice: <float name=\"gamma\" value=\"1.0\"/>");
        }
        mtsamd_material m;
        if (nestedBsdf->tag == "ref") {
            if (!named || !named->count(nestedBsdf->get("id"))) throw Fail("reference to unknown bsdf '" + nestedBsdf->get("id") + "'");
            m = copyMaterial(out, named->at(nestedBsdf->get("id")));
        } else m = makeBsdf(*nestedBsdf, opts, out, named);
        if (m.type == MTSAMD_BSDF_MIXTURE || m.two_sided || m.frame_kind)
            throw Fail(type + ": the nested BSDF must be a plain model (not a mixturebsdf, twosided, bumpmap or normalmap)");
        m.frame_kind = type == "bumpmap" ? MTSAMD_FRAME_BUMPMAP : MTSAMD_FRAME_NORMALMAP;
        m.frame_texture = (int32_t) (texture->tag == "ref" ? out.namedTextures.at(texture->get("id")) : makeTexture(*texture, out));
        return m;
    }
    if (type == "mixturebsdf") {                // src/bsdfs/mixturebsdf.cpp:66-85,279-287
        const Props mp = parseProps(node);
        std::vector<double> weights;
        {
            std::string text = getS(mp, "weights", "");
            for (char &ch : text) if (ch == ',') ch = ' ';
            std::istringstream in(text);
            std::string tok;
            while (in >> tok) {
                char *end = nullptr;
                const double w = strtod(tok.c_str(), &end);
                if (*end != '\0') throw Fail("Could not parse the BSDF weights!");
                if (w < 0) throw Fail("Invalid BSDF weight!");
                weights.push_back(w);
            }
        }
        std::vector<int32_t> children;
        for (const auto &c : node.children) {
            if (c->tag == "bsdf") {
                children.push_back((int32_t) addMaterial(out, makeBsdf(*c, opts, out, named)));
            } else if (c->tag == "ref") {
                if (!named || !named->count(c->get("id"))) throw Fail("reference to unknown bsdf '" + c->get("id") + "'");
                children.push_back((int32_t) addMaterial(out, copyMaterial(out, named->at(c->get("id")))));
            }
        }
        for (int32_t c : children) {
            const mtsamd_material &cm = out.materials[(size_t) c];
            if (cm.two_sided || cm.type == MTSAMD_BSDF_MIXTURE)
                throw Fail("mixturebsdf: twosided and mixturebsdf cannot be nested");
            if (cm.frame_kind) throw Fail("mixturebsdf: bumpmap and normalmap cannot be nested");
        }
        if (children.size() != weights.size())
            throw Fail("BSDF count mismatch: " + std::to_string(children.size()) + " bsdfs, but specified " + std::to_string(weights.size()) + " weights");
        if (children.empty() || children.size() > 4) throw Fail("mixturebsdf: between one and four nested BSDFs are supported");
        mtsamd_material m;
        memset(&m, 0, sizeof(m));
        m.type = MTSAMD_BSDF_MIXTURE;
        m.ensure_energy_conservation = getB(mp, "ensureEnergyConservation", true) ? 1 : 0;
        m.back_material = -1;
        m.mix_count = (int32_t) children.size();
        for (size_t i = 0; i < children.size(); ++i) { m.mix_material[i] = children[i]; m.mix_weight[i] = (float) weights[i]; }
        return m;
    }
    for (const auto &c : node.children)
        if (c->tag == "bsdf")
            throw Fail("nested <" + c->tag + "> inside <bsdf type=\"" + type + "\"> is not supported");
    const Props p = parseProps(node);
    mtsamd_material m;
    memset(&m, 0, sizeof(m));
    m.sample_visible = getB(p, "sampleVisible", true) ? 1 : 0;
    m.ensure_energy_conservation = getB(p, "ensureEnergyConservation", true) ? 1 : 0;
    const std::string distr = lower(getS(p, "distribution", "beckmann"));
    auto distribution = [&]() {
        if (distr == "beckmann") return (int) MTSAMD_DISTR_BECKMANN;
        if (distr == "ggx") return (int) MTSAMD_DISTR_GGX;
        if (distr == "phong" || distr == "as") throw Fail("the Phong/Ashikhmin-Shirley microfacet distribution is not supported");
        throw Fail("Specified an invalid distribution \"" + distr + "\", must be \"beckmann\" or \"ggx\"");
    };
    const double alpha = getF(p, "alpha", 0.1);
    const double alphaU = p.count("alpha") ? alpha : getF(p, "alphaU", 0.1), alphaV = p.count("alpha") ? alpha : getF(p, "alphaV", 0.1);
    m.eta[0] = m.eta[1] = m.eta[2] = 0; m.k[0] = m.k[1] = m.k[2] = 1; m.ext_eta = 1.000277f;
    m.int_ior = 1.5046f; m.ext_ior = 1.000277f;
    m.back_material = -1;
    for (int i = 0; i < 3; ++i) { m.diffuse_reflectance[i] = 0.5f; m.specular_reflectance[i] = 1.0f; m.specular_transmittance[i] = 1.0f; }

    if (type == "diffuse") {                    // src/bsdfs/diffuse.cpp:84-91
        m.type = MTSAMD_BSDF_DIFFUSE;
        getRgb(p, p.count("reflectance") ? "reflectance" : "diffuseReflectance", 0.5f, 0.5f, 0.5f, m.diffuse_reflectance);
    } else if (type == "dielectric") {          // src/bsdfs/dielectric.cpp:147-165
        m.type = MTSAMD_BSDF_DIELECTRIC;
        m.int_ior = lookupIor(p, "intIOR", "bk7"); m.ext_ior = lookupIor(p, "extIOR", "air");
        getRgb(p, "specularReflectance", 1, 1, 1, m.specular_reflectance);
        getRgb(p, "specularTransmittance", 1, 1, 1, m.specular_transmittance);
    } else if (type == "ward") {                // src/bsdfs/ward.cpp:97-126
        m.type = MTSAMD_BSDF_WARD;
        const std::string v = lower(getS(p, "variant", "balanced"));
        if (v == "ward") m.variant = 0; else if (v == "ward-duer") m.variant = 1; else if (v == "balanced") m.variant = 2;
        else throw Fail("Specified an invalid model type \"" + v + "\", must be \"ward\", \"ward-duer\", or \"balanced\"!");
        m.alpha_u = (float) getF(p, "alphaU", alpha); m.alpha_v = (float) getF(p, "alphaV", alpha);
        getRgb(p, "diffuseReflectance", 0.5f, 0.5f, 0.5f, m.diffuse_reflectance);
        getRgb(p, "specularReflectance", 0.2f, 0.2f, 0.2f, m.specular_reflectance);
    } else if (type == "roughconductor") {      // src/bsdfs/roughconductor.cpp:168-201
        m.type = MTSAMD_BSDF_ROUGHCONDUCTOR;
        m.distribution = distribution();
        m.alpha_u = (float) alphaU; m.alpha_v = (float) alphaV;
        conductorConstants(p, opts, m);
        m.ext_eta = lookupIor(p, "extEta", "air");
        getRgb(p, "specularReflectance", 1, 1, 1, m.specular_reflectance);
    } else if (type == "roughplastic") {        // src/bsdfs/roughplastic.cpp:197-228,281-299
        m.type = MTSAMD_BSDF_ROUGHPLASTIC;
        m.distribution = distribution();
        m.alpha_u = m.alpha_v = (float) alphaU;
        m.int_ior = lookupIor(p, "intIOR", "polypropylene"); m.ext_ior = lookupIor(p, "extIOR", "air");
        m.variant = getB(p, "nonlinear", false) ? 1 : 0;
        getRgb(p, "diffuseReflectance", 0.5f, 0.5f, 0.5f, m.diffuse_reflectance);
        getRgb(p, "specularReflectance", 1, 1, 1, m.specular_reflectance);
        if (opts.dataDir.empty()) throw Fail("roughplastic needs the rough transmittance tables: pass the directory that holds microfacet/" + distr + ".dat");
        std::vector<uint8_t> dat;
        const std::string path = opts.dataDir + "/microfacet/" + distr + ".dat";
        if (!readFile(path, dat)) throw Fail("cannot read " + path);
        bool texturedAlpha = false;     // <texture name="alpha"> or <ref name="alpha">: the roughness is looked up per intersection
        for (const auto &c : node.children) texturedAlpha |= (c->tag == "texture" || c->tag == "ref") && c->get("name") == "alpha";
        uint32_t n = 0;
        if (texturedAlpha) {
            // the table stays a function of the roughness (RoughPlastic::configure without setAlpha, roughplastic.cpp:281-299)
            out.rtransTables.emplace_back(1);
            (void) mtsamd_rtrans_reduce_eta(dat.data(), dat.size(), m.int_ior / m.ext_ior, out.rtransTables.back().data(), 0, &n);    // asks for the size
            out.rtransTables.back().assign(std::max<uint32_t>(n, 1), 0.0f);
            if (n == 0 || mtsamd_rtrans_reduce_eta(dat.data(), dat.size(), m.int_ior / m.ext_ior, out.rtransTables.back().data(), n, &n) != MTSAMD_OK)
                throw Fail("cannot reduce " + path + " to the material's index of refraction");
        } else {
            out.rtransTables.emplace_back(512);
            if (mtsamd_rtrans_reduce(dat.data(), dat.size(), m.int_ior / m.ext_ior, m.alpha_u, out.rtransTables.back().data(), 512, &n,
                                     &m.rtrans_diffuse_ext, &m.rtrans_diffuse_int) != MTSAMD_OK)
                throw Fail(mtsamd_last_error());
        }
        m.rtrans_n = n;     // the pointer is set in desc(), once the table storage no longer moves
    } else if (type == "roughdielectric") {     // src/bsdfs/roughdielectric.cpp:183-211
        m.type = MTSAMD_BSDF_ROUGHDIELECTRIC;
        m.distribution = distribution();
        m.alpha_u = (float) alphaU; m.alpha_v = (float) alphaV;
        m.int_ior = lookupIor(p, "intIOR", "bk7"); m.ext_ior = lookupIor(p, "extIOR", "air");
        getRgb(p, "specularReflectance", 1, 1, 1, m.specular_reflectance);
        getRgb(p, "specularTransmittance", 1, 1, 1, m.specular_transmittance);
    } else if (type == "conductor") {           // src/bsdfs/conductor.cpp:153-176
        m.type = MTSAMD_BSDF_CONDUCTOR;
        conductorConstants(p, opts, m);
        m.ext_eta = lookupIor(p, "extEta", "air");
        getRgb(p, "specularReflectance", 1, 1, 1, m.specular_reflectance);
    } else if (type == "plastic") {             // src/bsdfs/plastic.cpp:146-166
        m.type = MTSAMD_BSDF_PLASTIC;
        m.int_ior = lookupIor(p, "intIOR", "polypropylene"); m.ext_ior = lookupIor(p, "extIOR", "air");
        m.variant = getB(p, "nonlinear", false) ? 1 : 0;
        getRgb(p, "diffuseReflectance", 0.5f, 0.5f, 0.5f, m.diffuse_reflectance);
        getRgb(p, "specularReflectance", 1, 1, 1, m.specular_reflectance);
    } else {
        throw Fail("BSDF type '" + type + "' is outside the path hot path of this core");
    }
    attachTextures(node, out, m);
    return m;
}

// ---- meshes -----------------------------------------------------------------------------------------------------------------
// Point / Normal transforms as Transform::operator() applies them (include/mitsuba/core/transform.h), single precision.
void applyToWorld(Mesh &mesh, const Mat4 &md) {
    if (md.isIdentity()) return;
    float m[16];
    for (int i = 0; i < 16; ++i) m[i] = (float) md.m[i];
    const size_t nv = mesh.positions.size() / 3;
    for (size_t v = 0; v < nv; ++v) {
        float *p = &mesh.positions[3 * v];
        const float x = m[0] * p[0] + m[1] * p[1] + m[2] * p[2] + m[3];
        const float y = m[4] * p[0] + m[5] * p[1] + m[6] * p[2] + m[7];
        const float z = m[8] * p[0] + m[9] * p[1] + m[10] * p[2] + m[11];
        const float w = m[12] * p[0] + m[13] * p[1] + m[14] * p[2] + m[15];
        if (w == 1.0f) { p[0] = x; p[1] = y; p[2] = z; } else { p[0] = x / w; p[1] = y / w; p[2] = z / w; }
    }
    if (!mesh.normals.empty()) {
        Mat4 invd;
        if (!invert(md, invd)) throw Fail("singular toWorld transform");
        float inv[16];
        for (int i = 0; i < 16; ++i) inv[i] = (float) invd.m[i];
        for (size_t v = 0; v < nv; ++v) {       // inverse transpose, then normalize (src/shapes/serialized.cpp:184-202)
            float *n = &mesh.normals[3 * v];
            const float x = inv[0] * n[0] + inv[4] * n[1] + inv[8] * n[2];
            const float y = inv[1] * n[0] + inv[5] * n[1] + inv[9] * n[2];
            const float z = inv[2] * n[0] + inv[6] * n[1] + inv[10] * n[2];
            const float len = std::sqrt(x * x + y * y + z * z);
            n[0] = x / len; n[1] = y / len; n[2] = z / len;
        }
    }
    const double *a = md.m;
    const double det = a[0] * (a[5] * a[10] - a[6] * a[9]) - a[1] * (a[4] * a[10] - a[6] * a[8]) + a[2] * (a[4] * a[9] - a[5] * a[8]);
    if (det < 0)
        for (size_t t = 0; t + 2 < mesh.indices.size(); t += 3) std::swap(mesh.indices[t], mesh.indices[t + 1]);
}

// TriMesh::rebuildTopology (src/librender/trimesh.cpp:468-606): vertex normals are dropped and the vertices re-made so
// that, around every distinct (position, uv), the faces whose normals lie within maxAngle of the first unassigned
// face's normal share a vertex (greedy, in the order the faces name the vertex). Corners sharper than that are split.
void rebuildTopology(Mesh &mesh, float maxAngleDeg) {
    const size_t nt = mesh.indices.size() / 3;
    const bool hasUV = !mesh.texcoords.empty();
    const float dpThresh = std::cos(maxAngleDeg * (3.14159265358979323846f / 180.0f));
    struct Entry { float key[5]; uint32_t tri; bool clustered; };
    std::vector<Entry> entries;
    entries.reserve(nt * 3);
    std::vector<float> faceNormals(nt * 3);
    const float *P = mesh.positions.data();
    for (size_t t = 0; t < nt; ++t) {
        const uint32_t *idx = &mesh.indices[3 * t];
        for (int j = 0; j < 3; ++j) {
            Entry e;
            for (int k = 0; k < 3; ++k) e.key[k] = P[3 * (size_t) idx[j] + k];
            e.key[3] = hasUV ? mesh.texcoords[2 * (size_t) idx[j]] : 0.0f;
            e.key[4] = hasUV ? mesh.texcoords[2 * (size_t) idx[j] + 1] : 0.0f;
            e.tri = (uint32_t) t; e.clustered = false;
            entries.push_back(e);
        }
        const float *v0 = P + 3 * (size_t) idx[0], *v1 = P + 3 * (size_t) idx[1], *v2 = P + 3 * (size_t) idx[2];
        const float a[3] = { v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2] }, b[3] = { v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2] };
        float n[3] = { a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0] };
        const float l = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        if (l > 2.93873587705571876e-39f) { n[0] /= l; n[1] /= l; n[2] /= l; } else { n[0] = n[1] = n[2] = 0.0f; }     // degenerate triangle
        memcpy(&faceNormals[3 * t], n, 12);
    }
    // the order of a multimap keyed by (position, uv): ascending keys, equal keys in insertion order
    auto less = [](const Entry &x, const Entry &y) {
        for (int k = 0; k < 5; ++k) { if (x.key[k] < y.key[k]) return true; if (x.key[k] > y.key[k]) return false; }
        return false;
    };
    std::stable_sort(entries.begin(), entries.end(), less);
    std::vector<float> newPositions, newTexcoords;
    std::vector<uint32_t> newIndices(nt * 3, 0xFFFFFFFFu);
    for (size_t start = 0; start < entries.size();) {
        size_t end = start + 1;
        while (end < entries.size() && !less(entries[start], entries[end])) ++end;
        for (size_t i2 = start; i2 < end; ++i2) {
            if (entries[i2].clustered) continue;
            const float *n1 = &faceNormals[3 * (size_t) entries[i2].tri];
            const uint32_t vertexIdx = (uint32_t) (newPositions.size() / 3);
            newPositions.insert(newPositions.end(), entries[i2].key, entries[i2].key + 3);
            if (hasUV) newTexcoords.insert(newTexcoords.end(), entries[i2].key + 3, entries[i2].key + 5);
            for (size_t i3 = i2; i3 < end; ++i3) {
                Entry &t2 = entries[i3];
                if (t2.clustered) continue;
                const float *n2 = &faceNormals[3 * (size_t) t2.tri];
                const bool same = n1[0] == n2[0] && n1[1] == n2[1] && n1[2] == n2[2];
                if (same || n1[0] * n2[0] + n1[1] * n2[1] + n1[2] * n2[2] > dpThresh) {
                    const uint32_t *idx = &mesh.indices[3 * (size_t) t2.tri];
                    for (int k = 0; k < 3; ++k) {
                        const float *q = P + 3 * (size_t) idx[k];
                        if (q[0] == entries[i2].key[0] && q[1] == entries[i2].key[1] && q[2] == entries[i2].key[2])
                            newIndices[3 * (size_t) t2.tri + k] = vertexIdx;
                    }
                    t2.clustered = true;
                }
            }
        }
        start = end;
    }
    for (uint32_t v : newIndices) if (v == 0xFFFFFFFFu) throw Fail("rebuildTopology: a triangle corner was left without a vertex");
    mesh.indices.swap(newIndices);
    mesh.positions.swap(newPositions);
    if (hasUV) mesh.texcoords.swap(newTexcoords);
    mesh.normals.clear();
}

// TriMesh::computeNormals (src/librender/trimesh.cpp:636-672): angle-weighted face normals
void computeNormals(Mesh &mesh, bool flip) {
    const size_t nv = mesh.positions.size() / 3, nt = mesh.indices.size() / 3;
    mesh.normals.assign(nv * 3, 0.0f);
    const float *P = mesh.positions.data();
    for (size_t t = 0; t < nt; ++t) {
        const uint32_t *idx = &mesh.indices[3 * t];
        float n[3] = { 0, 0, 0 };
        for (int i = 0; i < 3; ++i) {
            const float *v0 = P + 3 * (size_t) idx[i], *v1 = P + 3 * (size_t) idx[(i + 1) % 3], *v2 = P + 3 * (size_t) idx[(i + 2) % 3];
            float a[3] = { v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2] }, b[3] = { v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2] };
            if (i == 0) {
                n[0] = a[1] * b[2] - a[2] * b[1]; n[1] = a[2] * b[0] - a[0] * b[2]; n[2] = a[0] * b[1] - a[1] * b[0];
                const float len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
                if (len == 0) break;
                for (float &c : n) c /= len;
            }
            const float la = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]), lb = std::sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
            for (int k = 0; k < 3; ++k) { a[k] /= la; b[k] /= lb; }
            // unitAngle (include/mitsuba/core/util.h:305-310)
            const float d = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
            float angle;
            if (d < 0) {
                const float s[3] = { b[0] + a[0], b[1] + a[1], b[2] + a[2] };
                angle = 3.14159265358979323846f - 2 * std::asin(0.5f * std::sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]));
            } else {
                const float s[3] = { b[0] - a[0], b[1] - a[1], b[2] - a[2] };
                angle = 2 * std::asin(0.5f * std::sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]));
            }
            for (int k = 0; k < 3; ++k) mesh.normals[3 * (size_t) idx[i] + k] += n[k] * angle;
        }
    }
    for (size_t v = 0; v < nv; ++v) {
        float *n = &mesh.normals[3 * v];
        float len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        if (flip) len *= -1;
        if (len != 0) { n[0] /= len; n[1] /= len; n[2] /= len; } else { n[0] = 1; n[1] = 0; n[2] = 0; }
    }
}

uint32_t addMaterial(LoadedScene &out, const mtsamd_material &m) { out.materials.push_back(m); return (uint32_t) out.materials.size() - 1; }

void addMesh(LoadedScene &out, const Mesh &mesh, uint32_t material, const float *radiance, float samplingWeight) {
    const uint32_t base = (uint32_t) (out.positions.size() / 3);
    const size_t nv = mesh.positions.size() / 3;
    mtsamd_shape s;
    memset(&s, 0, sizeof(s));
    s.first_triangle = (uint32_t) (out.indices.size() / 3);
    s.n_triangles = (uint32_t) (mesh.indices.size() / 3);
    s.material = (int32_t) material;
    s.emitter = -1;
    s.flags = (mesh.normals.empty() ? 0u : (uint32_t) MTSAMD_SHAPE_HAS_NORMALS) | (mesh.texcoords.empty() ? 0u : (uint32_t) MTSAMD_SHAPE_HAS_TEXCOORDS);
    for (uint32_t i : mesh.indices) {
        if (i >= nv) throw Fail("mesh '" + mesh.name + "' has a vertex index out of range");
        out.indices.push_back(base + i);
    }
    out.positions.insert(out.positions.end(), mesh.positions.begin(), mesh.positions.end());
    // per-vertex arrays are global: pad them for meshes that lack the attribute
    out.normals.resize(3 * (size_t) base, 0.0f);
    out.texcoords.resize(2 * (size_t) base, 0.0f);
    if (!mesh.normals.empty()) { out.normals.insert(out.normals.end(), mesh.normals.begin(), mesh.normals.end()); out.anyNormals = true; }
    if (!mesh.texcoords.empty()) { out.texcoords.insert(out.texcoords.end(), mesh.texcoords.begin(), mesh.texcoords.end()); out.anyTexcoords = true; }
    if (radiance) {
        mtsamd_emitter e;
        memset(&e, 0, sizeof(e));
        e.shape = (int32_t) out.shapes.size();
        memcpy(e.radiance, radiance, 12);
        e.sampling_weight = samplingWeight;
        s.emitter = (int32_t) out.emitters.size();
        out.emitters.push_back(e);
    }
    out.shapes.push_back(s);
}

// GaussianFilter / BoxFilter tables come from the library so that both hosts use the same values
void makeFilter(const Element *rf, mtsamd_film &film) {
    int rc;
    if (!rf || rf->get("type") == "gaussian") {
        rc = mtsamd_filter_gaussian(rf ? (float) getF(parseProps(*rf), "stddev", 0.5) : 0.5f, &film.filter_radius, film.filter_values);
    } else if (rf->get("type") == "box") {
        rc = mtsamd_filter_box((float) getF(parseProps(*rf), "radius", 0.5), &film.filter_radius, film.filter_values);
    } else if (rf->get("type") == "tent") {
        rc = mtsamd_filter_table(MTSAMD_FILTER_TENT, 0, 0, &film.filter_radius, film.filter_values);
    } else if (rf->get("type") == "mitchell") {
        const Props fp = parseProps(*rf);
        rc = mtsamd_filter_table(MTSAMD_FILTER_MITCHELL, (float) getF(fp, "B", 1.0 / 3.0), (float) getF(fp, "C", 1.0 / 3.0), &film.filter_radius, film.filter_values);
    } else if (rf->get("type") == "catmullrom") {
        rc = mtsamd_filter_table(MTSAMD_FILTER_CATMULLROM, 0, 0, &film.filter_radius, film.filter_values);
    } else if (rf->get("type") == "lanczos") {
        rc = mtsamd_filter_table(MTSAMD_FILTER_LANCZOS, (float) getF(parseProps(*rf), "lobes", 3), 0, &film.filter_radius, film.filter_values);
    } else {
        throw Fail("reconstruction filter '" + rf->get("type") + "' is not supported");
    }
    if (rc != MTSAMD_OK) throw Fail("invalid reconstruction filter parameters");
}

// WavefrontOBJ::addMaterial (src/shapes/obj.cpp:518-573): the BSDF an MTL material stands for. The Phong model (illum 2
// with a specular colour and an exponent) and opacity masks are BSDFs this core does not have: such a material is an
// error unless the scene gives the shape a <bsdf name="..."> of its own for it.
mtsamd_material makeMtlMaterial(const std::string &name, const MtlMaterial &mm, const std::string &dir, const LoadOptions &opts, LoadedScene &out) {
    if (mm.masked) throw Fail("OBJ material '" + name + "' has an opacity mask (the 'mask' BSDF is not supported); give the shape a <bsdf name=\"" + name + "\">");
    int model = mm.illum;
    const bool noSpecular = mm.mapKs.empty() && mm.ks[0] == 0 && mm.ks[1] == 0 && mm.ks[2] == 0;
    if (model == 2 && (noSpecular || mm.ns == 0)) model = 1;
    if (model == 2) throw Fail("OBJ material '" + name + "' uses the Phong model (illum 2), which is not supported; give the shape a <bsdf name=\"" + name + "\">");
    Element e;
    e.tag = "bsdf";
    auto child = [&](Element &parent, const char *tag, const char *pname, const std::string &value) -> Element & {
        parent.children.emplace_back(new Element());
        Element &c = *parent.children.back();
        c.tag = tag; c.attrs["name"] = pname; c.attrs["value"] = value;
        return c;
    };
    const std::string savedDir = out.sceneDir;
    out.sceneDir = dir;         // file names in the library are relative to the OBJ file (obj.cpp:229)
    mtsamd_material m;
    try {
        if (model == 4 || model == 6 || model == 7 || model == 9) e.attrs["type"] = "dielectric";
        else if (model == 5 || model == 8) { e.attrs["type"] = "conductor"; child(e, "string", "material", "Al"); }
        else {
            e.attrs["type"] = "diffuse";
            if (mm.hasMapKd) {
                e.children.emplace_back(new Element());
                Element &t = *e.children.back();
                t.tag = "texture"; t.attrs["type"] = "bitmap"; t.attrs["name"] = "reflectance";
                child(t, "string", "filename", mm.mapKd);
            } else {    // Spectrum::fromSRGB
                char buf[96];
                snprintf(buf, sizeof(buf), "%.9g, %.9g, %.9g", mm.kd[0], mm.kd[1], mm.kd[2]);
                child(e, "srgb", "reflectance", buf);
            }
        }
        m = makeBsdf(e, opts, out);
        if (!mm.bump.empty()) {     // obj.cpp:551-560: a bitmap read without a gamma curve
            Element t;
            t.tag = "texture"; t.attrs["type"] = "bitmap";
            child(t, "string", "filename", mm.bump);
            child(t, "float", "gamma", "1");
            m.frame_kind = MTSAMD_FRAME_BUMPMAP;
            m.frame_texture = (int32_t) makeTexture(t, out);
        }
    } catch (...) {
        out.sceneDir = savedDir;
        throw;
    }
    out.sceneDir = savedDir;
    return m;
}

}  // namespace

// ---- public pieces -----------------------------------------------------------------------------------------------------------
mtsamd_scene_desc LoadedScene::desc() const {
    mtsamd_scene_desc d;
    memset(&d, 0, sizeof(d));
    d.n_vertices = (uint32_t) (positions.size() / 3); d.n_triangles = (uint32_t) (indices.size() / 3);
    d.n_shapes = (uint32_t) shapes.size(); d.n_materials = (uint32_t) materials.size(); d.n_emitters = (uint32_t) emitters.size();
    d.positions = positions.data(); d.indices = indices.data();
    // pad the per-vertex attribute arrays up to the vertex count (meshes added last may lack them)
    LoadedScene *self = const_cast<LoadedScene *>(this);
    if (anyNormals) { self->normals.resize(positions.size(), 0.0f); d.normals = normals.data(); }
    if (anyTexcoords) { self->texcoords.resize(positions.size() / 3 * 2, 0.0f); d.texcoords = texcoords.data(); }
    size_t table = 0;
    for (mtsamd_material &m : self->materials)
        if (m.type == MTSAMD_BSDF_ROUGHPLASTIC) m.rtrans = rtransTables[table++].data();
    for (size_t i = 0; i < textures.size(); ++i)
        self->textures[i].pixels = texturePixels[i].empty() ? nullptr : texturePixels[i].data();
    d.n_textures = (uint32_t) textures.size(); d.textures = textures.empty() ? nullptr : textures.data();
    for (size_t i = 0; i < emitters.size() && i < emitterPixels.size(); ++i)
        if (!emitterPixels[i].empty()) self->emitters[i].pixels = emitterPixels[i].data();
    d.shapes = shapes.data(); d.materials = materials.data(); d.emitters = emitters.empty() ? nullptr : emitters.data();
    d.camera = camera; d.film = film; d.integrator = integrator;
    return d;
}

// TriMesh::loadCompressed (src/librender/trimesh.cpp:175-252) with the offset dictionary of :272-294
std::string loadSerialized(const std::string &path, int shapeIndex, Mesh &out) {
    std::vector<uint8_t> data;
    if (!readFile(path, data)) return "cannot read " + path;
    if (data.size() < 8) return "Encountered an invalid file format!";
    uint16_t fmt, version;
    memcpy(&fmt, &data[0], 2); memcpy(&version, &data[2], 2);
    if (fmt != 0x041C) return "Encountered an invalid file format!";
    if (version != 3 && version != 4) return "Encountered an incompatible file version!";
    size_t offset = 0;
    if (shapeIndex != 0) {
        uint32_t count;
        memcpy(&count, &data[data.size() - 4], 4);
        if (shapeIndex < 0 || (uint32_t) shapeIndex > count) return "Unable to unserialize mesh, shape index is out of range!";
        if (version == 4) {
            uint64_t o; const size_t at = data.size() - 8 * (size_t) (count - shapeIndex) - 4;
            if (at + 8 > data.size()) return "corrupt offset dictionary";
            memcpy(&o, &data[at], 8); offset = (size_t) o;
        } else {
            uint32_t o; const size_t at = data.size() - 4 * (size_t) (count - shapeIndex + 1);
            if (at + 4 > data.size()) return "corrupt offset dictionary";
            memcpy(&o, &data[at], 4); offset = o;
        }
        if (offset + 4 > data.size()) return "corrupt offset dictionary";
    }
    // inflate one zlib stream that starts after the 4-byte header of the chosen mesh
    std::vector<uint8_t> raw;
    {
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (inflateInit(&zs) != Z_OK) return "zlib initialisation failed";
        zs.next_in = data.data() + offset + 4;
        zs.avail_in = (uInt) std::min<size_t>(data.size() - offset - 4, 0xFFFFFFFFu);
        uint8_t buf[1 << 16];
        int rc;
        do {
            zs.next_out = buf; zs.avail_out = sizeof(buf);
            rc = inflate(&zs, Z_NO_FLUSH);
            if (rc != Z_OK && rc != Z_STREAM_END) { inflateEnd(&zs); return "inflate(): stream error"; }
            raw.insert(raw.end(), buf, buf + (sizeof(buf) - zs.avail_out));
        } while (rc != Z_STREAM_END);
        inflateEnd(&zs);
    }
    size_t p = 0;
    auto need = [&](size_t n) { if (p + n > raw.size()) throw Fail("truncated mesh stream"); };
    try {
        need(4);
        uint32_t flags; memcpy(&flags, &raw[p], 4); p += 4;
        if (version == 4) {
            size_t e = p;
            while (e < raw.size() && raw[e] != 0) ++e;
            out.name.assign((const char *) &raw[p], e - p);
            p = e + 1;
        }
        need(16);
        uint64_t nv, nt;
        memcpy(&nv, &raw[p], 8); memcpy(&nt, &raw[p + 8], 8); p += 16;
        const bool dbl = (flags & 0x2000u) != 0;
        // the counts come from the file: checked against the stream's length before anything is multiplied or allocated
        if (nv > raw.size() || nt > raw.size()) throw Fail("truncated mesh stream");
        auto take = [&](size_t n, int comps, std::vector<float> *dst) {
            const size_t count = n * comps, bytes = count * (dbl ? 8 : 4);
            need(bytes);
            if (dst && count) {
                dst->resize(count);
                if (dbl) for (size_t i = 0; i < count; ++i) { double v; memcpy(&v, &raw[p + 8 * i], 8); (*dst)[i] = (float) v; }
                else memcpy(dst->data(), &raw[p], bytes);
            }
            p += bytes;
        };
        take(nv, 3, &out.positions);
        if (flags & 0x1u) take(nv, 3, &out.normals);
        if (flags & 0x2u) take(nv, 2, &out.texcoords);
        if (flags & 0x8u) take(nv, 3, nullptr);
        out.faceNormals = (flags & 0x10u) != 0;
        need(nt * 12);
        out.indices.resize(nt * 3);
        if (nt) memcpy(out.indices.data(), &raw[p], nt * 12);
    } catch (const Fail &f) {
        return f.what();
    }
    return "";
}

// Wavefront OBJ (src/shapes/obj.cpp:244-335,371-391): v / vn / vt / f, polygons as fans, negative indices not used by the
// reference either; vertices are made unique per (position, texcoord, normal) index triple. Groups and materials are
// ignored: the file becomes one mesh.
// WavefrontOBJ (src/shapes/obj.cpp:224-334): the file is cut into one mesh per group ("g") and per material change
// ("usemtl") unless `collapse` is set; every mesh remembers the material that was current when it was closed.
std::string loadObjParts(const std::string &path, bool flipTexCoords, bool collapse, std::vector<ObjPart> &parts, std::string &materialLibrary) {
    std::ifstream f(path);
    if (!f) return "cannot read " + path;
    std::vector<float> vp, vn, vt;
    struct Corner { int p, t, n; };
    std::vector<Corner> corners;
    std::string materialName, error;
    auto trimmed = [](const std::string &s) {
        const size_t b = s.find_first_not_of(" \t\r\n"), e = s.find_last_not_of(" \t\r\n");
        return b == std::string::npos ? std::string() : s.substr(b, e - b + 1);
    };
    auto flush = [&]() {        // createMesh (obj.cpp:611-714)
        if (corners.empty()) return true;
        ObjPart part;
        part.material = materialName;
        Mesh &out = part.mesh;
        std::map<std::tuple<int, int, int>, uint32_t> unique;
        bool anyN = false, anyT = false, allN = true, allT = true;
        for (Corner &c : corners) {
            if (c.p < 0) c.p += (int) (vp.size() / 3) + 1;
            if (c.n < 0) c.n += (int) (vn.size() / 3) + 1;
            if (c.t < 0) c.t += (int) (vt.size() / 2) + 1;
            if (c.p <= 0 || (size_t) c.p > vp.size() / 3) { error = "OBJ face references a vertex that does not exist"; return false; }
            if (c.n) { anyN = true; if ((size_t) c.n > vn.size() / 3 || c.n < 0) { error = "OBJ face references a normal that does not exist"; return false; } } else allN = false;
            if (c.t) { anyT = true; if ((size_t) c.t > vt.size() / 2 || c.t < 0) { error = "OBJ face references a texture coordinate that does not exist"; return false; } } else allT = false;
        }
        const bool useN = anyN && allN, useT = anyT && allT;
        for (const Corner &c : corners) {
            const auto key = std::make_tuple(c.p, useT ? c.t : 0, useN ? c.n : 0);
            auto it = unique.find(key);
            if (it == unique.end()) {
                const uint32_t idx = (uint32_t) (out.positions.size() / 3);
                it = unique.emplace(key, idx).first;
                out.positions.insert(out.positions.end(), &vp[3 * (c.p - 1)], &vp[3 * (c.p - 1)] + 3);
                if (useN) out.normals.insert(out.normals.end(), &vn[3 * (c.n - 1)], &vn[3 * (c.n - 1)] + 3);
                if (useT) out.texcoords.insert(out.texcoords.end(), &vt[2 * (c.t - 1)], &vt[2 * (c.t - 1)] + 2);
            }
            out.indices.push_back(it->second);
        }
        parts.push_back(std::move(part));
        corners.clear();
        return true;
    };
    std::string line;
    while (std::getline(f, line)) {
        while (!line.empty() && line.back() == '\\') { std::string next; if (!std::getline(f, next)) break; line.pop_back(); line += next; }
        std::istringstream is(line);
        std::string tag;
        if (!(is >> tag)) continue;
        if (tag == "v") { float x, y, z; is >> x >> y >> z; vp.insert(vp.end(), { x, y, z }); }
        else if (tag == "vn") { float x, y, z; is >> x >> y >> z; vn.insert(vn.end(), { x, y, z }); }
        else if (tag == "vt") { float u, v; is >> u >> v; if (flipTexCoords) v = 1 - v; vt.insert(vt.end(), { u, v }); }
        else if (tag == "g" && !collapse) { if (!flush()) return error; }
        else if (tag == "usemtl") {
            if (!collapse && !flush()) return error;
            materialName = trimmed(line.substr(line.find("usemtl") + 6));
        }
        else if (tag == "mtllib") materialLibrary = trimmed(line.substr(line.find("mtllib") + 6));
        else if (tag == "f") {
            std::vector<Corner> poly;
            std::string tok;
            while (is >> tok) {
                Corner c = { 0, 0, 0 };
                const size_t s1 = tok.find('/');
                c.p = std::atoi(tok.substr(0, s1).c_str());
                if (s1 != std::string::npos) {
                    const size_t s2 = tok.find('/', s1 + 1);
                    if (s2 == std::string::npos) c.t = std::atoi(tok.substr(s1 + 1).c_str());
                    else { if (s2 > s1 + 1) c.t = std::atoi(tok.substr(s1 + 1, s2 - s1 - 1).c_str()); c.n = std::atoi(tok.substr(s2 + 1).c_str()); }
                }
                poly.push_back(c);
            }
            if (poly.size() < 3) return "Invalid OBJ face format!";
            for (size_t i = 2; i < poly.size(); ++i) { corners.push_back(poly[0]); corners.push_back(poly[i - 1]); corners.push_back(poly[i]); }
        }
    }
    if (!flush()) return error;
    if (parts.empty()) return "OBJ file has no faces";
    return "";
}

// The whole file as one mesh (what `collapse` gives)
std::string loadObj(const std::string &path, bool flipTexCoords, Mesh &out) {
    std::vector<ObjPart> parts;
    std::string lib;
    const std::string err = loadObjParts(path, flipTexCoords, true, parts, lib);
    if (!err.empty()) return err;
    out = std::move(parts[0].mesh);
    return "";
}

// A material library (WavefrontOBJ::loadMaterialLibrary, obj.cpp:423-505): the statements the reference reads
std::string loadMtl(const std::string &path, std::map<std::string, MtlMaterial> &out) {
    std::ifstream f(path);
    if (!f) return "";      // "Could not find referenced material library": a warning in the reference, no materials
    std::string line, current;
    while (std::getline(f, line)) {
        std::istringstream is(line);
        std::string tag;
        if (!(is >> tag)) continue;
        if (tag == "newmtl") {
            const size_t b = line.find("newmtl") + 6;
            current = line.substr(b);
            const size_t s0 = current.find_first_not_of(" \t\r\n"), s1 = current.find_last_not_of(" \t\r\n");
            current = s0 == std::string::npos ? std::string() : current.substr(s0, s1 - s0 + 1);
            out[current] = MtlMaterial();
        } else if (current.empty() && !out.count(current)) {
            continue;
        } else {
            MtlMaterial &m = out[current];
            if (tag == "Kd") is >> m.kd[0] >> m.kd[1] >> m.kd[2];
            else if (tag == "Ks") is >> m.ks[0] >> m.ks[1] >> m.ks[2];
            else if (tag == "Ns") is >> m.ns;
            else if (tag == "illum") is >> m.illum;
            else if (tag == "map_Kd") { is >> m.mapKd; m.hasMapKd = true; }
            else if (tag == "map_Ks") { is >> m.mapKs; }
            else if (tag == "bump") { is >> m.bump; }
            else if (tag == "map_d") { m.masked = true; }
            else if (tag == "d") { float v = 1; is >> v; m.masked = v != 1; }
        }
    }
    return "";
}

// The 24-vertex, 12-triangle unit cube of the `cube` plugin (src/shapes/cube.cpp:24-31): every face has its own four
// vertices with the face normal and the (0,1) (1,1) (1,0) (0,0) texture coordinates.
void makeCube(Mesh &out) {
    static const signed char P[24][3] = {
        { 1, -1, -1 }, { 1, -1, 1 }, { -1, -1, 1 }, { -1, -1, -1 }, { 1, 1, -1 }, { -1, 1, -1 }, { -1, 1, 1 }, { 1, 1, 1 },
        { 1, -1, -1 }, { 1, 1, -1 }, { 1, 1, 1 }, { 1, -1, 1 }, { 1, -1, 1 }, { 1, 1, 1 }, { -1, 1, 1 }, { -1, -1, 1 },
        { -1, -1, 1 }, { -1, 1, 1 }, { -1, 1, -1 }, { -1, -1, -1 }, { 1, 1, -1 }, { 1, -1, -1 }, { -1, -1, -1 }, { -1, 1, -1 } };
    static const signed char N[6][3] = { { 0, -1, 0 }, { 0, 1, 0 }, { 1, 0, 0 }, { 0, 0, 1 }, { -1, 0, 0 }, { 0, 0, -1 } };
    static const float UV[4][2] = { { 0, 1 }, { 1, 1 }, { 1, 0 }, { 0, 0 } };
    out.name = "cube";
    for (int v = 0; v < 24; ++v) {
        for (int k = 0; k < 3; ++k) { out.positions.push_back(P[v][k]); out.normals.push_back(N[v / 4][k]); }
        out.texcoords.push_back(UV[v % 4][0]); out.texcoords.push_back(UV[v % 4][1]);
    }
    for (uint32_t f = 0; f < 6; ++f) {
        const uint32_t b = 4 * f;
        const uint32_t tri[6] = { b, b + 1, b + 2, b + 3, b, b + 2 };
        out.indices.insert(out.indices.end(), tri, tri + 6);
    }
}

// The `rectangle` shape (src/shapes/rectangle.cpp:80-110,150-165): the square [-1, 1]^2 of the xy plane, normal +z,
// uv = (x + 1) / 2, (y + 1) / 2, as two triangles with face normals. The same surface, area and uv; the shading frame
// follows dp/du like the reference's (for a mirroring toWorld or flipNormals its second tangent points the other way,
// which only an anisotropic BSDF would notice).
void makeRectangle(Mesh &out) {
    static const float P[4][2] = { { -1, -1 }, { 1, -1 }, { 1, 1 }, { -1, 1 } };
    out.name = "rectangle";
    out.faceNormals = true;
    for (int v = 0; v < 4; ++v) {
        out.positions.push_back(P[v][0]); out.positions.push_back(P[v][1]); out.positions.push_back(0.0f);
        out.texcoords.push_back(0.5f * (P[v][0] + 1)); out.texcoords.push_back(0.5f * (P[v][1] + 1));
    }
    const uint32_t tri[6] = { 0, 1, 2, 2, 3, 0 };
    out.indices.assign(tri, tri + 6);
}

std::string writePfm(const std::string &path, const float *rgba, int width, int height) {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return "cannot write " + path;
    fprintf(f, "PF\n%d %d\n-1\n", width, height);
    std::vector<float> row((size_t) width * 3);
    for (int y = height - 1; y >= 0; --y) {
        for (int x = 0; x < width; ++x)
            for (int c = 0; c < 3; ++c) row[3 * (size_t) x + c] = rgba[4 * ((size_t) y * width + x) + c];
        fwrite(row.data(), sizeof(float), row.size(), f);
    }
    fclose(f);
    return "";
}

// PNG: signature, IHDR, one IDAT holding the zlib stream of the unfiltered rows, IEND
std::string writePng(const std::string &path, const uint8_t *pixels, int width, int height, int channels, const std::vector<Metadata> *metadata) {
    if (channels < 1 || channels > 4) return "PNG output needs 1 to 4 channels";
    const size_t row = (size_t) width * channels;
    std::vector<uint8_t> raw((row + 1) * (size_t) height);
    for (int y = 0; y < height; ++y) {
        raw[(row + 1) * y] = 0;     // filter type "none"
        memcpy(&raw[(row + 1) * y + 1], pixels + row * y, row);
    }
    uLongf zlen = compressBound((uLong) raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong) raw.size(), 6) != Z_OK) return "zlib compression failed";
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return "cannot write " + path;
    auto be32 = [](uint8_t *p, uint32_t v) { p[0] = (uint8_t) (v >> 24); p[1] = (uint8_t) (v >> 16); p[2] = (uint8_t) (v >> 8); p[3] = (uint8_t) v; };
    auto chunk = [&](const char *type, const uint8_t *data, size_t n) {
        uint8_t head[8], tail[4];
        be32(head, (uint32_t) n);
        memcpy(head + 4, type, 4);
        uLong crc = crc32(0L, head + 4, 4);
        if (n) crc = crc32(crc, data, (uInt) n);
        be32(tail, (uint32_t) crc);
        fwrite(head, 1, 8, f);
        if (n) fwrite(data, 1, n, f);
        fwrite(tail, 1, 4, f);
    };
    const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    fwrite(sig, 1, 8, f);
    uint8_t ihdr[13];
    be32(ihdr, (uint32_t) width); be32(ihdr + 4, (uint32_t) height);
    const uint8_t colourType[5] = { 0, 0, 4, 2, 6 };    // grey, grey + alpha, RGB, RGBA
    ihdr[8] = 8; ihdr[9] = colourType[channels]; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    chunk("IHDR", ihdr, 13);
    if (metadata)
        for (const Metadata &m : *metadata) {       // keyword, a zero byte, the text
            std::ostringstream value;
            if (m.kind == 0) value << m.s; else if (m.kind == 1) value << m.i; else value << m.f;
            std::string body = m.key.substr(0, 79);
            body.push_back('\0');
            body += value.str();
            chunk("tEXt", (const uint8_t *) body.data(), body.size());
        }
    chunk("IDAT", z.data(), zlen);
    chunk("IEND", nullptr, 0);
    const bool ok = !ferror(f);
    fclose(f);
    return ok ? "" : "cannot write " + path;
}

// OpenEXR, the format HDRFilm writes by default (src/films/hdrfilm.cpp:213-223, Bitmap::writeOpenEXR in
// src/libcore/bitmap.cpp): here a single-part scan-line file, version 2, three 32-bit float channels (B, G, R in the
// file's alphabetical order), no compression, rows top to bottom. Any OpenEXR reader opens it; the reference's
// default of half-precision samples with PIZ compression is a storage choice, not a different image.
// float -> half, round to nearest even, overflow to infinity (what OpenEXR's half(float) does)
static uint16_t floatToHalf(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t) (sign | 0x7c00u | (x > 0x7f800000u ? 0x200u | ((x >> 13) & 0x3ffu) : 0u));     // inf, NaN
    if (x >= 0x477ff000u) return (uint16_t) (sign | 0x7c00u);       // rounds to 65520 or more: infinity
    if (x < 0x33000001u) return (uint16_t) sign;                     // rounds to zero (at most half the smallest subnormal)
    if (x < 0x38800000u) {                                           // subnormal half
        const int shift = 126 - (int) (x >> 23);                     // 14..24
        const uint32_t mant = (x & 0x7fffffu) | 0x800000u;
        uint32_t h = mant >> shift;
        const uint32_t rem = mant & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (h & 1u))) ++h;
        return (uint16_t) (sign | h);
    }
    uint32_t h = (x - 0x38000000u) >> 13;
    const uint32_t rem = x & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;
    return (uint16_t) (sign | h);
}

std::string writeExr(const std::string &path, const float *rgba, int width, int height, int pixelFormat, bool half, const std::vector<Metadata> *metadata) {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return "cannot write " + path;
    std::vector<uint8_t> h;
    auto put = [&](const void *p, size_t n) { h.insert(h.end(), (const uint8_t *) p, (const uint8_t *) p + n); };
    auto putStr = [&](const char *z) { put(z, strlen(z) + 1); };
    auto putI = [&](int32_t v) { put(&v, 4); };
    auto putF = [&](float v) { put(&v, 4); };
    auto attr = [&](const char *name, const char *type, int32_t size) { putStr(name); putStr(type); putI(size); };
    // channels in the alphabetical order the format wants; source: 0..3 = R, G, B, alpha of the developed image, 4 = luminance
    struct Channel { const char *name; int source; };
    std::vector<Channel> channels;
    const bool colour = pixelFormat == MTSAMD_PIXEL_RGB || pixelFormat == MTSAMD_PIXEL_RGBA;
    const bool alpha = pixelFormat == MTSAMD_PIXEL_LUMINANCE_ALPHA || pixelFormat == MTSAMD_PIXEL_RGBA;
    if (alpha) channels.push_back({ "A", 3 });
    if (colour) { channels.push_back({ "B", 2 }); channels.push_back({ "G", 1 }); channels.push_back({ "R", 0 }); }
    else channels.push_back({ "Y", 4 });
    const size_t nc = channels.size(), bpc = half ? 2 : 4;
    bool longNames = false;         // attribute names above 31 characters need the long-names bit of the version field
    if (metadata) for (const Metadata &m : *metadata) longNames |= m.key.size() > 31;
    putI(20000630); putI(2 | (longNames ? 0x400 : 0));
    attr("channels", "chlist", (int32_t) (nc * 18 + 1));
    for (const Channel &c : channels) { putStr(c.name); putI(half ? 1 /* HALF */ : 2 /* FLOAT */); putI(0); putI(1); putI(1); }
    h.push_back(0);
    attr("compression", "compression", 1); h.push_back(0);
    attr("dataWindow", "box2i", 16); putI(0); putI(0); putI(width - 1); putI(height - 1);
    attr("displayWindow", "box2i", 16); putI(0); putI(0); putI(width - 1); putI(height - 1);
    attr("lineOrder", "lineOrder", 1); h.push_back(0);
    attr("pixelAspectRatio", "float", 4); putF(1.0f);
    attr("screenWindowCenter", "v2f", 8); putF(0.0f); putF(0.0f);
    attr("screenWindowWidth", "float", 4); putF(1.0f);
    if (metadata)
        for (const Metadata &m : *metadata) {
            if (m.key.empty() || m.key.size() > 255) { fclose(f); return "metadata key '" + m.key + "' cannot be an OpenEXR attribute name"; }
            if (m.kind == 0) { attr(m.key.c_str(), "string", (int32_t) m.s.size()); put(m.s.data(), m.s.size()); }
            else if (m.kind == 1) { attr(m.key.c_str(), "int", 4); putI(m.i); }
            else { attr(m.key.c_str(), "float", 4); putF(m.f); }
        }
    h.push_back(0);
    const uint64_t dataBytes = (uint64_t) width * nc * bpc, rowBytes = 8 + dataBytes;
    uint64_t offset = h.size() + (uint64_t) height * 8;
    for (int y = 0; y < height; ++y, offset += rowBytes) put(&offset, 8);
    fwrite(h.data(), 1, h.size(), f);
    std::vector<uint8_t> row(dataBytes);
    for (int y = 0; y < height; ++y) {
        const int32_t head[2] = { y, (int32_t) dataBytes };
        fwrite(head, 4, 2, f);
        for (size_t c = 0; c < nc; ++c)         // one plane per channel within the row
            for (int x = 0; x < width; ++x) {
                const float *px = rgba + 4 * ((size_t) y * width + x);
                // Spectrum::getLuminance for RGB (src/libcore/spectrum.cpp)
                const float v = channels[c].source == 4 ? px[0] * 0.212671f + px[1] * 0.715160f + px[2] * 0.072169f : px[channels[c].source];
                uint8_t *dst = &row[(c * (size_t) width + x) * bpc];
                if (half) { const uint16_t hv = floatToHalf(v); memcpy(dst, &hv, 2); } else memcpy(dst, &v, 4);
            }
        fwrite(row.data(), 1, row.size(), f);
    }
    fclose(f);
    return "";
}

// timeString / memString (src/libcore/util.cpp:864-905)
static std::string timeString(double seconds, bool precise) {
    float time = (float) seconds;
    if (std::isnan(time) || std::isinf(time)) return "inf";
    char suffix = 's';
    if (time > 60) { time /= 60; suffix = 'm'; if (time > 60) { time /= 60; suffix = 'h'; if (time > 12) { time /= 12; suffix = 'd'; } } }
    std::ostringstream os;
    os << std::setprecision(precise ? 4 : 1) << std::fixed << time << suffix;
    return os.str();
}
static std::string memString(size_t size, bool precise) {
    float value = (float) size;
    const char *suffixes[] = { "B", "KiB", "MiB", "GiB", "TiB", "PiB" };
    int suffix = 0;
    while (suffix < 5 && value > 1024.0f) { value /= 1024.0f; ++suffix; }
    std::ostringstream os;
    os << std::setprecision(suffix == 0 ? 0 : (precise ? 4 : 1)) << std::fixed << value << " " << suffixes[suffix];
    return os.str();
}
static size_t residentBytes() {     // what getPrivateMemoryUsage reports on Linux: the resident set
    FILE *f = fopen("/proc/self/statm", "r");
    if (!f) return 0;
    unsigned long total = 0, resident = 0;
    const int n = fscanf(f, "%lu %lu", &total, &resident);
    fclose(f);
    return n == 2 ? (size_t) resident * 4096u : 0;
}

std::string resolveAnnotations(const LoadedScene &scene, const std::string &destFile, double renderSeconds, int coreCount, std::vector<Metadata> &out) {
    out.clear();
    bool generatedBy = false;
    auto strip = [](std::string v) { v.erase(std::remove_if(v.begin(), v.end(), [](unsigned char c) { return std::isspace(c); }), v.end()); return v; };
    for (const LoadedScene::Annotation &a : scene.annotations) {
        Metadata m;
        m.key = a.key; m.kind = a.kind; m.i = (int32_t) a.i; m.f = (float) a.f;
        if (a.kind == 0) {
            std::string value = a.text;
            while (true) {      // annotations.h:76-141: replace the first $source['key'] until none is left
                const size_t strt = value.find('$');
                if (strt == std::string::npos) break;
                const size_t par1 = value.find('[', strt);
                if (par1 == std::string::npos) break;
                const size_t par2 = value.find(']', par1);
                if (par2 == std::string::npos) break;
                const std::string source = strip(value.substr(strt + 1, par1 - strt - 1));
                std::string key = strip(value.substr(par1 + 1, par2 - par1 - 1));
                if (key.size() < 2 || key.front() != '\'' || key.back() != '\'') return "Encountered invalid key '" + key + "'";
                key = key.substr(1, key.size() - 2);
                std::string replacement;
                if (source == "scene") {
                    if (key == "renderTime") replacement = timeString(renderSeconds, false);
                    else if (key == "renderTimePrecise") replacement = timeString(renderSeconds, true);
                    else if (key == "memUsage") replacement = memString(residentBytes(), false);
                    else if (key == "memUsagePrecise") replacement = memString(residentBytes(), true);
                    else if (key == "coreCount") replacement = std::to_string(coreCount);
                    else if (key == "blockSize") replacement = std::to_string(scene.blockSize);
                    else if (key == "sourceFile") replacement = scene.sourceFile;
                    else if (key == "destFile") replacement = destFile;
                    else return "Property \"" + key + "\" has not been specified!";
                } else {
                    auto el = scene.elements.find(source);
                    if (source != "film" && source != "sampler" && source != "sensor" && source != "integrator")
                        return "Unknown data source '" + source + "' (must be film/sampler/sensor/integrator)";
                    if (el == scene.elements.end() || !el->second.count(key)) return "Property \"" + key + "\" has not been specified!";
                    replacement = el->second.at(key);
                }
                value.replace(strt, par2 - strt + 1, replacement);
            }
            m.s = value;
        }
        generatedBy |= m.key == "generatedBy";
        out.push_back(m);
    }
    if (!generatedBy) {     // Bitmap::write adds it (bitmap.cpp:2610-2611, 3192-3193); this caller names itself
        Metadata m;
        m.key = "generatedBy"; m.s = "mtsamd-render (libmtsamd, interface version " + std::to_string(MTSAMD_ABI_VERSION) + ")";
        out.push_back(m);
    }
    std::sort(out.begin(), out.end(), [](const Metadata &a, const Metadata &b) { return a.key < b.key; });     // Properties keeps its names ordered
    return "";
}

// Stanford PLY (src/shapes/ply.cpp:91-137,202-316): ASCII and both binary byte orders; vertex properties x y z,
// nx ny nz, u|texture_u|s and v|texture_v|t (anything else, vertex colours included, is skipped); faces are
// triangles or quads through vertex_indices|vertex_index, a quad (a,b,c,d) becoming (a,b,c) and (d,a,c).
// Positions and normals stay in object space here; the caller applies toWorld.
namespace {
struct PlyProperty { std::string name; int type = 0, countType = 0; bool isList = false; };
struct PlyElement { std::string name; size_t count = 0; std::vector<PlyProperty> props; };

// type codes: 1 int8, 2 uint8, 3 int16, 4 uint16, 5 int32, 6 uint32, 7 float32, 8 float64
int plyType(const std::string &t) {
    static const std::pair<const char *, int> names[] = {
        { "char", 1 }, { "int8", 1 }, { "uchar", 2 }, { "uint8", 2 }, { "short", 3 }, { "int16", 3 }, { "ushort", 4 }, { "uint16", 4 },
        { "int", 5 }, { "int32", 5 }, { "uint", 6 }, { "uint32", 6 }, { "float", 7 }, { "float32", 7 }, { "double", 8 }, { "float64", 8 } };
    for (const auto &n : names) if (t == n.first) return n.second;
    throw Fail("PLY: unknown property type '" + t + "'");
}
int plySize(int type) { static const int size[] = { 0, 1, 1, 2, 2, 4, 4, 4, 8 }; return size[type]; }

struct PlyReader {
    const std::vector<uint8_t> &data;
    size_t pos;
    int format;     // 0 ascii, 1 little endian, 2 big endian
    double next(int type) {
        if (format == 0) {
            while (pos < data.size() && std::isspace(data[pos])) ++pos;
            const size_t b = pos;
            while (pos < data.size() && !std::isspace(data[pos])) ++pos;
            if (b == pos) throw Fail("PLY: unexpected end of file");
            return std::stod(std::string((const char *) &data[b], pos - b));
        }
        const int n = plySize(type);
        if (pos + n > data.size()) throw Fail("PLY: unexpected end of file");
        uint8_t raw[8];
        for (int i = 0; i < n; ++i) raw[i] = data[pos + (format == 1 ? i : n - 1 - i)];     // to little endian
        pos += n;
        switch (type) {
            case 1: { int8_t v; memcpy(&v, raw, 1); return v; }
            case 2: return raw[0];
            case 3: { int16_t v; memcpy(&v, raw, 2); return v; }
            case 4: { uint16_t v; memcpy(&v, raw, 2); return v; }
            case 5: { int32_t v; memcpy(&v, raw, 4); return v; }
            case 6: { uint32_t v; memcpy(&v, raw, 4); return v; }
            case 7: { float v; memcpy(&v, raw, 4); return v; }
            default: { double v; memcpy(&v, raw, 8); return v; }
        }
    }
};
}  // namespace

std::string loadPly(const std::string &path, Mesh &out) {
    try {
        std::vector<uint8_t> data;
        if (!readFile(path, data)) return "PLY file \"" + path + "\" could not be found!";
        // header: lines up to end_header
        size_t pos = 0;
        auto line = [&]() {
            const size_t b = pos;
            while (pos < data.size() && data[pos] != '\n') ++pos;
            std::string l((const char *) &data[b], pos - b);
            if (pos < data.size()) ++pos;
            if (!l.empty() && l.back() == '\r') l.pop_back();
            return l;
        };
        if (line() != "ply") return "\"" + path + "\" is not a PLY file";
        int format = -1;
        std::vector<PlyElement> elements;
        for (;;) {
            if (pos >= data.size()) return "PLY: header has no end_header";
            std::istringstream is(line());
            std::string key;
            is >> key;
            if (key == "end_header") break;
            if (key == "format") {
                std::string f; is >> f;
                format = f == "ascii" ? 0 : f == "binary_little_endian" ? 1 : f == "binary_big_endian" ? 2 : -1;
                if (format < 0) return "PLY: unknown format '" + f + "'";
            } else if (key == "element") {
                PlyElement e; is >> e.name >> e.count; elements.push_back(e);
            } else if (key == "property") {
                if (elements.empty()) return "PLY: property outside an element";
                PlyProperty pr; std::string t; is >> t;
                if (t == "list") { std::string ct, it; is >> ct >> it >> pr.name; pr.isList = true; pr.countType = plyType(ct); pr.type = plyType(it); }
                else { pr.type = plyType(t); is >> pr.name; }
                elements.back().props.push_back(pr);
            }       // comment, obj_info: ignored
        }
        if (format < 0) return "PLY: header has no format line";
        PlyReader rd{ data, pos, format };
        // every record of an element with properties takes at least one byte of the file: a count beyond that is damage
        for (const PlyElement &e : elements)
            if (!e.props.empty() && e.count > data.size()) return "PLY: element '" + e.name + "' claims more records than the file has bytes";
        out = Mesh();
        size_t slash = path.find_last_of('/');
        out.name = path.substr(slash == std::string::npos ? 0 : slash + 1);
        for (const PlyElement &e : elements) {
            if (e.name == "vertex") {
                int slot[8];    // x y z nx ny nz u v -> property index
                std::fill(slot, slot + 8, -1);
                for (size_t k = 0; k < e.props.size(); ++k) {
                    const std::string &n = e.props[k].name;
                    if (e.props[k].isList) continue;
                    if (n == "x") slot[0] = (int) k; else if (n == "y") slot[1] = (int) k; else if (n == "z") slot[2] = (int) k;
                    else if (n == "nx") slot[3] = (int) k; else if (n == "ny") slot[4] = (int) k; else if (n == "nz") slot[5] = (int) k;
                    else if (n == "u" || n == "texture_u" || n == "s") slot[6] = (int) k;
                    else if (n == "v" || n == "texture_v" || n == "t") slot[7] = (int) k;
                }
                if (slot[0] < 0 || slot[1] < 0 || slot[2] < 0) return "PLY: vertices have no x/y/z";
                const bool hasN = slot[3] >= 0, hasUV = slot[6] >= 0;       // the reference keys on nx and on u (ply.cpp:205-218)
                out.positions.resize(e.count * 3);
                if (hasN) out.normals.assign(e.count * 3, 0.0f);
                if (hasUV) out.texcoords.assign(e.count * 2, 0.0f);
                std::vector<double> row(e.props.size());
                for (size_t v = 0; v < e.count; ++v) {
                    for (size_t k = 0; k < e.props.size(); ++k) {
                        if (e.props[k].isList) { const size_t c = (size_t) rd.next(e.props[k].countType); for (size_t j = 0; j < c; ++j) rd.next(e.props[k].type); row[k] = 0; }
                        else row[k] = rd.next(e.props[k].type);
                    }
                    for (int c = 0; c < 3; ++c) out.positions[3 * v + c] = (float) row[slot[c]];
                    if (hasN) for (int c = 0; c < 3; ++c) if (slot[3 + c] >= 0) out.normals[3 * v + c] = (float) row[slot[3 + c]];
                    if (hasUV) for (int c = 0; c < 2; ++c) if (slot[6 + c] >= 0) out.texcoords[2 * v + c] = (float) row[slot[6 + c]];
                }
            } else if (e.name == "face") {
                const size_t nVerts = out.positions.size() / 3;
                for (size_t f = 0; f < e.count; ++f)
                    for (const PlyProperty &pr : e.props) {
                        if (!pr.isList) { rd.next(pr.type); continue; }
                        const size_t c = (size_t) rd.next(pr.countType);
                        const bool indices = pr.name == "vertex_indices" || pr.name == "vertex_index";
                        if (indices && c != 3 && c != 4)
                            return "Encountered a face with " + std::to_string(c) + " vertices! Only triangle and quad-based PLY meshes are supported for now.";
                        uint32_t face[4] = { 0, 0, 0, 0 };
                        for (size_t j = 0; j < c; ++j) {
                            const double v = rd.next(pr.type);
                            if (indices) {
                                if (v < 0 || (size_t) v >= nVerts) return "PLY: vertex index out of range";
                                face[j] = (uint32_t) v;
                            }
                        }
                        if (!indices) continue;
                        out.indices.insert(out.indices.end(), { face[0], face[1], face[2] });
                        if (c == 4) out.indices.insert(out.indices.end(), { face[3], face[0], face[2] });
                    }
            } else {
                for (size_t i = 0; i < e.count; ++i)
                    for (const PlyProperty &pr : e.props) {
                        if (pr.isList) { const size_t c = (size_t) rd.next(pr.countType); for (size_t j = 0; j < c; ++j) rd.next(pr.type); }
                        else rd.next(pr.type);
                    }
            }
        }
        if (out.indices.empty() || out.positions.empty()) return "Unable to load \"" + path + "\" (no triangles or vertices found)!";
        return "";
    } catch (const std::exception &e) {
        return e.what();
    }
}

namespace {
// SceneHandler::startElement (src/librender/scenehandler.cpp:207-223): "$name" in any attribute value is replaced by the
// parameter's value, longest-sorting names first; <default name= value=> defines a parameter unless it is already set
// (:684-688). Runs in document order, so a default only affects what follows it.
void substituteParameters(Element &el, std::map<std::string, std::string> &params) {
    for (auto &kv : el.attrs) {
        std::string &v = kv.second;
        if (v.find('$') == std::string::npos) continue;
        for (auto it = params.rbegin(); it != params.rend(); ++it) {
            const std::string needle = "$" + it->first;
            size_t pos = 0;
            while ((pos = v.find(needle, pos)) != std::string::npos) { v.replace(pos, needle.size(), it->second); ++pos; }
        }
        if (v.find('$') != std::string::npos && v.find('[') == std::string::npos)
            throw Fail("The scene referenced an undefined parameter: \"" + v + "\"");
    }
    if (el.tag == "default" && !params.count(el.get("name"))) params[el.get("name")] = el.get("value");
    for (auto &c : el.children) substituteParameters(*c, params);
}

// <include filename=.../> (scenehandler.cpp:658-682): the included file is a scene of its own whose children join the
// including scene; parameters and named objects are shared. Relative names resolve against the including file.
void expandIncludes(Element &scene, const std::string &base, std::map<std::string, std::string> &params, int depth) {
    if (depth > 16) throw Fail("<include> nesting is too deep");
    std::vector<std::unique_ptr<Element>> flat;
    for (auto &c : scene.children) {
        substituteParameters(*c, params);
        if (c->tag != "include") { flat.push_back(std::move(c)); continue; }
        const std::string file = joinPath(base, c->get("filename"));
        std::vector<uint8_t> bytes;
        if (!readFile(file, bytes)) throw Fail("cannot read included file " + file);
        const std::string text((const char *) bytes.data(), bytes.size());
        std::unique_ptr<Element> inc = xml_lite::Parser(text).parse();
        if (inc->tag != "scene") throw Fail("included file " + file + ": root element must be <scene>");
        expandIncludes(*inc, dirOf(file), params, depth + 1);
        for (auto &ic : inc->children) flat.push_back(std::move(ic));
    }
    scene.children = std::move(flat);
}
// The scalar properties of an element as Properties::getAsString prints them (src/libcore/properties.cpp:184-201): what
// $film['...'] and its siblings of a metadata annotation expand to; "type" is the plugin's name.
std::map<std::string, std::string> propertyStrings(const Element &el) {
    std::map<std::string, std::string> out;
    out["type"] = el.get("type");
    for (const auto &cp : el.children) {
        const Element &c = *cp;
        const std::string name = c.get("name");
        std::ostringstream oss;
        if (c.tag == "float") oss << (float) std::stod(c.get("value"));
        else if (c.tag == "integer") oss << std::stol(c.get("value"));
        else if (c.tag == "boolean") oss << (lower(c.get("value")) == "true" ? "true" : "false");
        else if (c.tag == "string") oss << c.get("value");
        else continue;
        if (name != "type") out[name] = oss.str();
    }
    return out;
}

// annotations.h:41-68,143-150: the film properties named metadata['key']; labels need a bitmap font this caller does not carry
void collectAnnotations(const Element &film, LoadedScene &out) {
    for (const auto &cp : film.children) {
        const Element &c = *cp;
        std::string key = c.get("name");
        key.erase(std::remove_if(key.begin(), key.end(), [](unsigned char ch) { return std::isspace(ch); }), key.end());
        const std::string lkey = lower(key);
        if (lkey.compare(0, 6, "label[") == 0 && lkey.back() == ']')
            throw Fail("label annotations (" + c.get("name") + ") are not supported: they are drawn with the renderer's built-in bitmap font; metadata['...'] annotations are");
        if (!(lkey.size() >= 12 && lkey.compare(0, 10, "metadata['") == 0 && lkey.compare(lkey.size() - 2, 2, "']") == 0)) continue;
        LoadedScene::Annotation a;
        a.key = key.substr(10, key.size() - 12);
        if (c.tag == "string") { a.kind = 0; a.text = c.get("value"); }
        else if (c.tag == "integer") { a.kind = 1; a.i = std::stol(c.get("value")); }
        else if (c.tag == "float") { a.kind = 2; a.f = std::stod(c.get("value")); }
        else if (c.tag == "boolean") { a.kind = 0; a.text = lower(c.get("value")) == "true" ? "true" : "false"; }
        else throw Fail("metadata annotation '" + a.key + "': only string, integer, float and boolean values are supported");
        out.annotations.push_back(a);
    }
}
}  // namespace

std::string loadSceneXml(const std::string &path, const LoadOptions &opts, LoadedScene &out) {
    try {
        std::vector<uint8_t> bytes;
        if (!readFile(path, bytes)) return "cannot read " + path;
        const std::string text((const char *) bytes.data(), bytes.size());
        std::unique_ptr<Element> root = xml_lite::Parser(text).parse();
        if (root->tag != "scene") return "root element must be <scene>";
        const std::string base = dirOf(path);
        std::map<std::string, std::string> params = opts.params;
        expandIncludes(*root, base, params, 0);
        std::map<std::string, uint32_t> namedBsdfs;
        out.sceneDir = base;
        out.sourceFile = path;
        g_spectrumDir = base;
        for (const auto &el : root->children)       // textures first: BSDFs refer to them by id
            if (el->tag == "texture") {
                const uint32_t idx = makeTexture(*el, out);
                if (el->has("id")) out.namedTextures[el->get("id")] = idx;
            }
        int defaultBsdf = -1;
        bool haveSensor = false;
        memset(&out.integrator, 0, sizeof(out.integrator));
        out.integrator.max_depth = -1; out.integrator.rr_depth = 5;
        out.integrator.emitter_samples = out.integrator.bsdf_samples = 1;

        auto bsdfIndex = [&](const Element &el) {
            const mtsamd_material made = makeBsdf(el, opts, out, &namedBsdfs);
            const uint32_t idx = addMaterial(out, made);
            if (el.has("id")) namedBsdfs[el.get("id")] = idx;
            return idx;
        };
        for (const auto &el : root->children) {
            if (el->tag == "bsdf") bsdfIndex(*el);
            else if (el->tag == "alias") {      // scenehandler.cpp:646-656
                const std::string id = el->get("id"), as = el->get("as");
                if (!namedBsdfs.count(id)) return "Referenced object '" + id + "' not found!";
                if (namedBsdfs.count(as)) return "Duplicate ID '" + as + "' used in scene description!";
                namedBsdfs[as] = namedBsdfs[id];
            }
        }

        for (const auto &elp : root->children) {
            const Element &el = *elp;
            if (el.tag == "integrator") {
                const Props p = parseProps(el);
                out.elements["integrator"] = propertyStrings(el);
                out.integratorType = el.get("type");
                out.integrator.max_depth = (int32_t) getF(p, "maxDepth", -1);
                out.integrator.rr_depth = (int32_t) getF(p, "rrDepth", 5);
                out.integrator.strict_normals = getB(p, "strictNormals", false) ? 1 : 0;
                out.integrator.hide_emitters = getB(p, "hideEmitters", false) ? 1 : 0;
                out.integrator.type = MTSAMD_INTEGRATOR_PATH;
                out.integrator.emitter_samples = out.integrator.bsdf_samples = 1;
                if (out.integratorType == "direct") {       // src/integrators/direct/direct.cpp:93-107
                    out.integrator.type = MTSAMD_INTEGRATOR_DIRECT;
                    const int shading = (int) getF(p, "shadingSamples", 1);
                    out.integrator.emitter_samples = (int32_t) getF(p, "emitterSamples", shading);
                    out.integrator.bsdf_samples = (int32_t) getF(p, "bsdfSamples", shading);
                } else if (out.integratorType == "volpath_simple") {       // media are not loaded: the surface-only form of it
                    out.integrator.type = MTSAMD_INTEGRATOR_VOLPATH_SIMPLE;
                }
            } else if (el.tag == "sensor") {
                if (el.get("type") != "perspective") return "only the perspective sensor is supported";
                const Props p = parseProps(el);
                out.elements["sensor"] = propertyStrings(el);
                int fw = 768, fh = 576;         // Film defaults (src/librender/film.cpp:25-55)
                int cropX = 0, cropY = 0, cropW = fw, cropH = fh;
                const Element *rf = nullptr;
                for (const auto &c : el.children) {
                    if (c->tag == "film") {
                        const Props fp = parseProps(*c);
                        out.elements["film"] = propertyStrings(*c);
                        collectAnnotations(*c, out);
                        fw = (int) getF(fp, "width", 768); fh = (int) getF(fp, "height", 576);
                        // crop window in pixels, by default the whole film (src/librender/film.cpp:35-48)
                        cropX = (int) getF(fp, "cropOffsetX", 0); cropY = (int) getF(fp, "cropOffsetY", 0);
                        cropW = (int) getF(fp, "cropWidth", fw); cropH = (int) getF(fp, "cropHeight", fh);
                        if (cropX < 0 || cropY < 0 || cropW <= 0 || cropH <= 0 || cropX + cropW > fw || cropY + cropH > fh)
                            return "Invalid crop window specification!";
                        for (const auto &r : c->children) if (r->tag == "rfilter") rf = r.get();
                        out.filmType = c->get("type");
                        if (out.filmType == "hdrfilm") {    // src/films/hdrfilm.cpp:213-260
                            const std::string pf = lower(getS(fp, "pixelFormat", "rgb")), cf = lower(getS(fp, "componentFormat", "float16"));
                            if (pf == "luminance") out.hdrPixelFormat = MTSAMD_PIXEL_LUMINANCE;
                            else if (pf == "luminancealpha") out.hdrPixelFormat = MTSAMD_PIXEL_LUMINANCE_ALPHA;
                            else if (pf == "rgb") out.hdrPixelFormat = MTSAMD_PIXEL_RGB;
                            else if (pf == "rgba") out.hdrPixelFormat = MTSAMD_PIXEL_RGBA;
                            else return "hdrfilm: pixel format '" + pf + "' is not supported (luminance, luminanceAlpha, rgb, rgba)";
                            if (cf == "float16") out.hdrHalf = true;
                            else if (cf == "float32") out.hdrHalf = false;
                            else return "hdrfilm: component format '" + cf + "' is not supported (float16, float32)";
                        }
                        if (out.filmType == "ldrfilm") {    // src/films/ldrfilm.cpp:131-183
                            const std::string fileFormat = lower(getS(fp, "fileFormat", "png"));
                            if (fileFormat == "jpg" || fileFormat == "jpeg") return "ldrfilm: JPEG output is not supported (use fileFormat=png)";
                            if (fileFormat != "png") return "The \"fileFormat\" parameter must either be equal to \"png\" or \"jpeg\"!";
                            const std::string pf = lower(getS(fp, "pixelFormat", "rgb")), tm = lower(getS(fp, "tonemapMethod", "gamma"));
                            if (pf == "luminance") out.ldr.pixel_format = MTSAMD_PIXEL_LUMINANCE;
                            else if (pf == "luminancealpha") out.ldr.pixel_format = MTSAMD_PIXEL_LUMINANCE_ALPHA;
                            else if (pf == "rgb") out.ldr.pixel_format = MTSAMD_PIXEL_RGB;
                            else if (pf == "rgba") out.ldr.pixel_format = MTSAMD_PIXEL_RGBA;
                            else return "The \"pixelFormat\" parameter must either be equal to \"luminance\", \"luminanceAlpha\", \"rgb\", or \"rgba\"!";
                            if (tm == "gamma") out.ldr.tonemap = MTSAMD_TONEMAP_GAMMA;
                            else if (tm == "reinhard") out.ldr.tonemap = MTSAMD_TONEMAP_REINHARD;
                            else return "The \"tonemapMethod\" parameter must either be equal to \"gamma\" or \"reinhard\"!";
                            out.ldr.gamma = (float) getF(fp, "gamma", -1);
                            out.ldr.exposure = (float) getF(fp, "exposure", 0.0);
                            out.ldr.key = (float) getF(fp, "key", 0.18);
                            out.ldr.burn = (float) getF(fp, "burn", 0.0);
                        }
                    } else if (c->tag == "sampler") {
                        out.elements["sampler"] = propertyStrings(*c);
                        out.sampleCount = (int) getF(parseProps(*c), "sampleCount", 4);
                    }
                }
                if (opts.width > 0 && opts.height > 0) { fw = opts.width; fh = opts.height; cropX = cropY = 0; cropW = fw; cropH = fh; }
                const double aspect = (double) fw / fh;
                // PerspectiveCamera::configure (src/librender/sensor.cpp:239-278): fov along an axis, or a focal length on
                // 35 mm film (36 x 24 mm, diagonal field of view); setYFov / setDiagonalFov (sensor.cpp:296-305)
                if (p.count("fov") && p.count("focalLength")) return "Please specify either a focal length ('focalLength') or a field of view ('fov')!";
                const double kPi = 3.14159265358979323846;
                double fov = getF(p, "fov", 0);
                std::string axis = lower(getS(p, "fovAxis", "x"));
                if (!p.count("fov")) {
                    std::string fl = getS(p, "focalLength", "50mm");
                    if (fl.size() >= 2 && fl.compare(fl.size() - 2, 2, "mm") == 0) fl = fl.substr(0, fl.size() - 2);
                    char *end = nullptr;
                    const double value = strtod(fl.c_str(), &end);
                    if (*end != '\0') return "Could not parse the focal length (must be of the form <x>mm, where <x> is a positive integer)!";
                    fov = 2 * 180 / kPi * std::atan(std::sqrt((double) (36 * 36 + 24 * 24)) / (2 * value));
                    axis = "diagonal";
                }
                if (axis == "smaller") axis = aspect > 1 ? "y" : "x";
                else if (axis == "larger") axis = aspect > 1 ? "x" : "y";
                if (axis == "y") {
                    fov = 2 * std::atan(std::tan(0.5 * fov * kPi / 180.0) * aspect) * 180.0 / kPi;
                } else if (axis == "diagonal") {
                    const double diagonal = 2 * std::tan(0.5 * fov * kPi / 180.0);
                    const double width = diagonal / std::sqrt(1.0 + 1.0 / (aspect * aspect));
                    fov = 2 * std::atan(width * 0.5) * 180.0 / kPi;
                } else if (axis != "x") {
                    return "The 'fovAxis' parameter must be set to one of 'smaller', 'larger', 'diagonal', 'x', or 'y'!";
                }
                if (fov <= 0 || fov >= 180) return "The horizontal field of view must be in the interval (0, 180)!";
                memset(&out.camera, 0, sizeof(out.camera));
                Mat4 tw = p.count("toWorld") ? p.at("toWorld").t : Mat4::identity();
                for (int i = 0; i < 16; ++i) out.camera.to_world[i] = (float) tw.m[i];
                out.camera.x_fov = (float) fov;
                out.camera.near_clip = (float) getF(p, "nearClip", 1e-2); out.camera.far_clip = (float) getF(p, "farClip", 1e4);
                memset(&out.film, 0, sizeof(out.film));
                out.film.width = fw; out.film.height = fh;
                out.film.crop_x = cropX; out.film.crop_y = cropY; out.film.crop_w = cropW; out.film.crop_h = cropH;
                makeFilter(rf, out.film);
                haveSensor = true;
            } else if (el.tag == "shape") {
                const Props p = parseProps(el);
                int mat = -1;
                std::map<std::string, int> namedMat;     // <bsdf name="..."> children: materials of an OBJ file by name (obj.cpp:736-757)
                bool emitter = false;
                float radiance[3] = { 1, 1, 1 }, weight = 1.0f;
                for (const auto &c : el.children) {
                    int made = -1;
                    if (c->tag == "ref") {
                        auto it = namedBsdfs.find(c->get("id"));
                        if (it == namedBsdfs.end()) return "reference to unknown id '" + c->get("id") + "'";
                        made = (int) it->second;
                    } else if (c->tag == "bsdf") {
                        made = (int) bsdfIndex(*c);
                    } else if (c->tag == "emitter") {
                        if (c->get("type") != "area") return "only area emitters are supported";
                        const Props ep = parseProps(*c);
                        getRgb(ep, "radiance", 1, 1, 1, radiance);
                        weight = (float) getF(ep, "samplingWeight", 1.0);
                        emitter = true;
                    } else if (c->tag == "medium" || c->tag == "subsurface") {
                        return "<" + c->tag + "> is not handled by the path integrator";
                    }
                    if (made >= 0) {
                        if (el.get("type") == "obj" && !c->get("name").empty()) namedMat[c->get("name")] = made;
                        else mat = made;
                    }
                }
                auto defaultMaterial = [&]() {      // Shape::configure attaches a default diffuse BSDF (src/librender/shape.cpp:48-53)
                    if (defaultBsdf < 0) {
                        Element e; e.tag = "bsdf"; e.attrs["type"] = "diffuse";
                        defaultBsdf = (int) addMaterial(out, makeBsdf(e, opts, out));
                    }
                    return defaultBsdf;
                };
                std::vector<std::pair<Mesh, int>> meshes;       // mesh, material
                const std::string type = el.get("type");
                std::string err;
                if (type == "obj") {
                    const std::string objPath = joinPath(base, getS(p, "filename", ""));
                    std::vector<ObjPart> parts;
                    std::string library;
                    err = loadObjParts(objPath, getB(p, "flipTexCoords", true), getB(p, "collapse", false), parts, library);
                    if (!err.empty()) return err;
                    std::map<std::string, MtlMaterial> mtl;
                    if (!library.empty() && getB(p, "loadMaterials", true)) {
                        err = loadMtl(joinPath(dirOf(objPath), library), mtl);
                        if (!err.empty()) return err;
                    }
                    std::map<std::string, int> fromLibrary;
                    const int shapeIndex = (int) getF(p, "shapeIndex", -1);
                    for (size_t k = 0; k < parts.size(); ++k) {
                        if (shapeIndex >= 0 && (int) k != shapeIndex) continue;
                        int m = -1;
                        const std::string &name = parts[k].material;
                        if (namedMat.count(name)) m = namedMat[name];
                        else if (mat >= 0) m = mat;
                        else if (mtl.count(name)) {
                            if (!fromLibrary.count(name)) fromLibrary[name] = (int) addMaterial(out, makeMtlMaterial(name, mtl[name], dirOf(objPath), opts, out));
                            m = fromLibrary[name];
                        } else m = defaultMaterial();
                        meshes.emplace_back(std::move(parts[k].mesh), m);
                    }
                    if (meshes.empty()) return "Unable to load the OBJ file: shape index is out of range";
                    if (emitter && meshes.size() > 1) return "Cannot attach an emitter to an OBJ file containing multiple objects!";
                } else {
                    Mesh mesh;
                    if (type == "serialized") err = loadSerialized(joinPath(base, getS(p, "filename", "")), (int) getF(p, "shapeIndex", 0), mesh);
                    else if (type == "ply") err = loadPly(joinPath(base, getS(p, "filename", "")), mesh);
                    else if (type == "cube") makeCube(mesh);
                    else if (type == "rectangle") makeRectangle(mesh);
                    else return "shape type '" + type + "' is not supported (triangle meshes only)";
                    if (!err.empty()) return err;
                    meshes.emplace_back(std::move(mesh), mat >= 0 ? mat : defaultMaterial());
                }
                for (auto &entry : meshes) {
                    Mesh &mesh = entry.first;
                    // whatever the file says: no triangle may name a vertex the mesh does not have
                    const size_t meshVertices = mesh.positions.size() / 3;
                    for (uint32_t index : mesh.indices)
                        if (index >= meshVertices) return "shape '" + type + "': a triangle refers to vertex " + std::to_string(index) + " of " + std::to_string(meshVertices);
                    if ((!mesh.normals.empty() && mesh.normals.size() != mesh.positions.size()) || (!mesh.texcoords.empty() && mesh.texcoords.size() / 2 != meshVertices))
                        return "shape '" + type + "': normals or texture coordinates do not match the vertex count";
                    const bool faceNormals = getB(p, "faceNormals", false) || mesh.faceNormals;
                    const bool flipNormals = getB(p, "flipNormals", false);
                    if (faceNormals) mesh.normals.clear();
                    applyToWorld(mesh, p.count("toWorld") ? p.at("toWorld").t : Mat4::identity());
                    if (p.count("maxSmoothAngle")) {        // serialized.cpp:204-209, ply.cpp:124-129, obj.cpp:336-343
                        if (faceNormals) return "The properties 'maxSmoothAngle' and 'faceNormals' can't be specified at the same time!";
                        rebuildTopology(mesh, (float) getF(p, "maxSmoothAngle", 0));
                    }
                    // TriMesh::computeNormals (src/librender/trimesh.cpp:606-672): flipNormals swaps the winding of a
                    // face-normal mesh, negates given vertex normals, and negates computed ones
                    if (faceNormals) {
                        if (flipNormals)
                            for (size_t t = 0; t + 2 < mesh.indices.size(); t += 3) std::swap(mesh.indices[t], mesh.indices[t + 1]);
                    } else if (!mesh.normals.empty()) {
                        if (flipNormals) for (float &c : mesh.normals) c *= -1;
                    } else {
                        computeNormals(mesh, flipNormals);
                    }
                    addMesh(out, mesh, (uint32_t) entry.second, emitter ? radiance : nullptr, weight);
                }
            } else if (el.tag == "bsdf") {
                // handled above
            } else if (el.tag == "emitter") {
                const std::string etype = el.get("type");
                const Props ep = parseProps(el);
                mtsamd_emitter e;
                memset(&e, 0, sizeof(e));
                e.shape = -1;
                e.sampling_weight = (float) getF(ep, "samplingWeight", 1.0);
                Mat4 toWorld = ep.count("toWorld") ? ep.at("toWorld").t : Mat4::identity();
                if (etype == "constant") {          // src/emitters/constant.cpp:47-50
                    getRgb(ep, "radiance", 1, 1, 1, e.radiance);
                } else if (etype == "envmap") {     // src/emitters/envmap.cpp:105-189
                    e.type = MTSAMD_EMITTER_ENVMAP;
                    if (!ep.count("filename")) return "envmap: the 'filename' parameter is missing";
                    if (ep.count("intensityScale")) return "The 'intensityScale' parameter has been deprecated and is now called scale.";
                    std::string file = getS(ep, "filename", "");
                    if (!file.empty() && file[0] != '/') file = out.sceneDir + "/" + file;
                    mtsamd_image img;
                    if (mtsamd_image_load(file.c_str(), (float) getF(ep, "gamma", 0.0), &img) != MTSAMD_OK) return mtsamd_last_error();
                    std::vector<float> px((size_t) img.width * img.height * 3);
                    for (size_t k = 0; k < (size_t) img.width * img.height; ++k)
                        for (int c = 0; c < 3; ++c) px[3 * k + c] = img.pixels[k * img.channels + (img.channels == 3 ? c : 0)];
                    e.width = img.width; e.height = img.height;
                    e.scale = (float) getF(ep, "scale", 1.0);
                    mtsamd_image_free(&img);
                    out.emitterPixels.resize(out.emitters.size() + 1);
                    out.emitterPixels.back() = std::move(px);
                } else if (etype == "point") {      // src/emitters/point.cpp:58-69
                    e.type = MTSAMD_EMITTER_POINT;
                    getRgb(ep, "intensity", 1, 1, 1, e.radiance);
                    if (ep.count("position")) {
                        if (ep.count("toWorld")) return "Only one of the parameters 'position' and 'toWorld' can be used!";
                        for (int i = 0; i < 3; ++i) toWorld.m[4 * i + 3] = ep.at("position").rgb[i];
                    }
                } else if (etype == "spot") {       // src/emitters/spot.cpp:67-78
                    e.type = MTSAMD_EMITTER_SPOT;
                    getRgb(ep, "intensity", 1, 1, 1, e.radiance);
                    for (const auto &cp : el.children)
                        if (cp->tag == "texture") return "spot emitter: only a constant texture is supported";
                    float tex[3];
                    getRgb(ep, "texture", 1, 1, 1, tex);
                    for (int i = 0; i < 3; ++i) e.radiance[i] *= tex[i];
                    e.cutoff_angle = (float) getF(ep, "cutoffAngle", 20);
                    e.beam_width = (float) getF(ep, "beamWidth", e.cutoff_angle * 3.0f / 4.0f);
                    if (!(e.cutoff_angle >= e.beam_width)) return "spot emitter: the cutoff angle must not be smaller than the beam width";
                } else if (etype == "directional") {    // src/emitters/directional.cpp:56-73
                    e.type = MTSAMD_EMITTER_DIRECTIONAL;
                    getRgb(ep, "irradiance", 1, 1, 1, e.radiance);
                    if (ep.count("direction")) {
                        if (ep.count("toWorld")) return "Only one of the parameters 'direction' and 'toWorld' can be used at a time!";
                        const float *v = ep.at("direction").rgb;
                        const float len = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
                        if (len == 0) return "directional emitter: the direction is zero";
                        const float a[3] = { v[0] / len, v[1] / len, v[2] / len };
                        float c[3];     // coordinateSystem (src/libcore/util.cpp:592-601): u = cross(c, a)
                        if (std::fabs(a[0]) > std::fabs(a[1])) {
                            const float inv = 1.0f / std::sqrt(a[0] * a[0] + a[2] * a[2]);
                            c[0] = a[2] * inv; c[1] = 0.0f; c[2] = -a[0] * inv;
                        } else {
                            const float inv = 1.0f / std::sqrt(a[1] * a[1] + a[2] * a[2]);
                            c[0] = 0.0f; c[1] = a[2] * inv; c[2] = -a[1] * inv;
                        }
                        const float u[3] = { c[1] * a[2] - c[2] * a[1], c[2] * a[0] - c[0] * a[2], c[0] * a[1] - c[1] * a[0] };
                        toWorld = lookAt({ 0, 0, 0 }, { a[0], a[1], a[2] }, { u[0], u[1], u[2] });
                    }
                } else {
                    return "emitter type '" + etype + "' is not supported (area, constant, envmap, point, spot, directional)";
                }
                for (int i = 0; i < 16; ++i) e.to_world[i] = (float) toWorld.m[i];
                out.emitters.push_back(e);
            } else if (el.tag == "default" || el.tag == "alias") {
                // handled above
            } else if (el.tag == "medium" || el.tag == "texture") {
                // a medium no shape refers to is inert; textures were read before the BSDFs
            }
        }
        if (!haveSensor) return "scene has no sensor";
        if (opts.maxDepth != 0) out.integrator.max_depth = opts.maxDepth;
        return "";
    } catch (const std::exception &e) {
        return e.what();
    }
}

}  // namespace mtsamd_host
