// mtsamd-render: standalone caller of the C ABI. Loads a Mitsuba 0.5 scene file (the subset scene_loader.h lists),
// renders it with the `path` integrator on one MI355X and writes a PFM or OpenEXR image, the role `mitsuba scene.xml -o out`
// plays for the reference (src/mitsuba/mitsuba.cpp:129-397, film output src/films/hdrfilm.cpp:481-537).
//
//   mtsamd-render scene.xml -o out.pfm|out.exr|out.png [--spp N] [--size W H] [--seed S] [--device D | --devices D0,D1,...]
//                 [--data-dir DIR] [--max-depth N] [--describe] [--film out.raw] [-D name=value ...]
//
// --devices renders one job on several GPUs of the node (mtsamd_render_multi: the scene is created once per GPU, the
// frame is dealt out as interleaved full-width tiles, the films are summed) -- the role of `mitsuba -p` worker threads.
// --describe prints a JSON summary of the loaded scene and exits without touching the GPU.
// --film also stores the unnormalised (R,G,B,alpha,weight) float32 film.
#include "scene_loader.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace mtsamd_host;

static int usage() {
    fprintf(stderr, "usage: mtsamd-render scene.xml -o out.pfm|out.exr|out.png [--spp N] [--size W H] [--seed S] [--device D | --devices D0,D1,...] [--data-dir DIR] "
                    "[--max-depth N] [--describe] [--film out.raw] [-D name=value ...]\n");
    return 2;
}

static void describe(const LoadedScene &s) {
    double sp = 0, sn = 0, st = 0;
    unsigned long long si = 0;
    for (float v : s.positions) sp += v;
    for (float v : s.normals) sn += v;
    for (float v : s.texcoords) st += v;
    for (uint32_t v : s.indices) si += v;
    printf("{\"vertices\": %zu, \"triangles\": %zu, \"shapes\": %zu, \"materials\": %zu, \"emitters\": %zu, \"has_normals\": %s, "
           "\"has_texcoords\": %s, \"sum_positions\": %.9g, \"sum_normals\": %.9g, \"sum_texcoords\": %.9g, \"sum_indices\": %llu, "
           "\"width\": %d, \"height\": %d, \"crop\": [%d, %d, %d, %d], \"x_fov\": %.9g, \"filter_radius\": %.9g, \"max_depth\": %d, \"rr_depth\": %d, "
           "\"sample_count\": %d, \"integrator\": \"%s\", \"integrator_kind\": %d, \"emitter_samples\": %d, \"bsdf_samples\": %d, \"to_world\": [",
           s.positions.size() / 3, s.indices.size() / 3, s.shapes.size(), s.materials.size(), s.emitters.size(),
           s.anyNormals ? "true" : "false", s.anyTexcoords ? "true" : "false", sp, sn, st, si, s.film.width, s.film.height,
           s.film.crop_x, s.film.crop_y, s.film.crop_w, s.film.crop_h, s.camera.x_fov, s.film.filter_radius, s.integrator.max_depth, s.integrator.rr_depth, s.sampleCount, s.integratorType.c_str(), s.integrator.type,
           s.integrator.emitter_samples, s.integrator.bsdf_samples);
    for (int i = 0; i < 16; ++i) printf("%s%.9g", i ? ", " : "", s.camera.to_world[i]);
    printf("], \"material_types\": [");
    for (size_t i = 0; i < s.materials.size(); ++i) printf("%s%d", i ? ", " : "", s.materials[i].type);
    printf("], \"two_sided\": [");
    for (size_t i = 0; i < s.materials.size(); ++i) printf("%s%d", i ? ", " : "", s.materials[i].two_sided);
    printf("], \"back_materials\": [");
    for (size_t i = 0; i < s.materials.size(); ++i) printf("%s%d", i ? ", " : "", s.materials[i].back_material);
    printf("], \"eta_k\": [");
    for (size_t i = 0; i < s.materials.size(); ++i) {
        const mtsamd_material &m = s.materials[i];
        printf("%s[%.9g, %.9g, %.9g, %.9g, %.9g, %.9g]", i ? ", " : "", m.eta[0], m.eta[1], m.eta[2], m.k[0], m.k[1], m.k[2]);
    }
    printf("], \"mixtures\": [");
    bool firstMix = true;
    for (size_t i = 0; i < s.materials.size(); ++i) {
        const mtsamd_material &m = s.materials[i];
        if (m.type != MTSAMD_BSDF_MIXTURE) continue;
        printf("%s{\"material\": %zu, \"children\": [", firstMix ? "" : ", ", i);
        for (int k = 0; k < m.mix_count; ++k) printf("%s%d", k ? ", " : "", m.mix_material[k]);
        printf("], \"weights\": [");
        for (int k = 0; k < m.mix_count; ++k) printf("%s%.9g", k ? ", " : "", m.mix_weight[k]);
        printf("]}");
        firstMix = false;
    }
    printf("], \"material_textures\": [");
    for (size_t i = 0; i < s.materials.size(); ++i)
        printf("%s[%d, %d, %d]", i ? ", " : "", s.materials[i].diffuse_texture, s.materials[i].specular_texture, s.materials[i].transmittance_texture);
    printf("], \"alpha_textures\": [");
    for (size_t i = 0; i < s.materials.size(); ++i) printf("%s[%d, %d]", i ? ", " : "", s.materials[i].alpha_u_texture, s.materials[i].alpha_v_texture);
    printf("], \"rtrans_tables\": [");       // per material: entries of its rough-transmittance table and their sum
    for (size_t i = 0, table = 0; i < s.materials.size(); ++i) {       // (tables belong to the roughplastic records, in their order)
        double sum = 0;
        uint32_t n = 0;
        if (s.materials[i].type == MTSAMD_BSDF_ROUGHPLASTIC && table < s.rtransTables.size()) {
            n = s.materials[i].rtrans_n;
            for (uint32_t k = 0; k < n && k < s.rtransTables[table].size(); ++k) sum += s.rtransTables[table][k];
            ++table;
        }
        printf("%s[%u, %.9g]", i ? ", " : "", n, sum);
    }
    printf("], \"envmaps\": [");
    bool firstEnv = true;
    for (size_t i = 0; i < s.emitters.size(); ++i) {
        const mtsamd_emitter &e = s.emitters[i];
        if (e.type != MTSAMD_EMITTER_ENVMAP) continue;
        double sum = 0;
        for (float v : s.emitterPixels[i]) sum += v;
        printf("%s{\"emitter\": %zu, \"size\": [%d, %d], \"scale\": %.9g, \"weight\": %.9g, \"sum_pixels\": %.9g, \"to_world\": [", firstEnv ? "" : ", ", i,
               e.width, e.height, e.scale, e.sampling_weight, sum);
        for (int k = 0; k < 16; ++k) printf("%s%.9g", k ? ", " : "", e.to_world[k]);
        printf("]}");
        firstEnv = false;
    }
    printf("], \"frames\": [");
    for (size_t i = 0; i < s.materials.size(); ++i) printf("%s[%d, %d]", i ? ", " : "", s.materials[i].frame_kind, s.materials[i].frame_texture);
    printf("], \"textures\": [");
    for (size_t i = 0; i < s.textures.size(); ++i) {
        const mtsamd_texture &t = s.textures[i];
        double sum = 0;
        for (float v : s.texturePixels[i]) sum += v;
        printf("%s{\"type\": %d, \"color0\": [%.9g, %.9g, %.9g], \"color1\": [%.9g, %.9g, %.9g], \"line_width\": %.9g, \"uv\": [%.9g, %.9g, %.9g, %.9g], "
               "\"scale\": %.9g, \"size\": [%d, %d, %d], \"filter_type\": %d, \"wrap\": [%d, %d], \"max_anisotropy\": %.9g, \"sum_pixels\": %.9g}",
               i ? ", " : "", t.type, t.color0[0], t.color0[1], t.color0[2], t.color1[0], t.color1[1], t.color1[2], t.line_width,
               t.u_offset, t.v_offset, t.u_scale, t.v_scale, t.scale, t.width, t.height, t.channels, t.filter_type, t.wrap_u, t.wrap_v,
               t.max_anisotropy, sum);
    }
    printf("], \"shape_triangles\": [");
    for (size_t i = 0; i < s.shapes.size(); ++i) printf("%s%u", i ? ", " : "", s.shapes[i].n_triangles);
    printf("], \"shape_materials\": [");
    for (size_t i = 0; i < s.shapes.size(); ++i) printf("%s%d", i ? ", " : "", s.shapes[i].material);
    printf("], \"diffuse\": [");
    for (size_t i = 0; i < s.materials.size(); ++i)
        printf("%s[%.9g, %.9g, %.9g]", i ? ", " : "", s.materials[i].diffuse_reflectance[0], s.materials[i].diffuse_reflectance[1], s.materials[i].diffuse_reflectance[2]);
    printf("], \"annotations\": [");
    for (size_t i = 0; i < s.annotations.size(); ++i) printf("%s[\"%s\", %d]", i ? ", " : "", s.annotations[i].key.c_str(), s.annotations[i].kind);
    printf("], \"emitter_records\": [");
    for (size_t i = 0; i < s.emitters.size(); ++i) {
        const mtsamd_emitter &e = s.emitters[i];
        printf("%s{\"type\": %d, \"shape\": %d, \"radiance\": [%.9g, %.9g, %.9g], \"sampling_weight\": %.9g, \"cutoff_angle\": %.9g, "
               "\"beam_width\": %.9g, \"to_world\": [", i ? ", " : "", e.type, e.shape, e.radiance[0], e.radiance[1], e.radiance[2],
               e.sampling_weight, e.cutoff_angle, e.beam_width);
        for (int k = 0; k < 16; ++k) printf("%s%.9g", k ? ", " : "", e.to_world[k]);
        printf("]}");
    }
    printf("]}\n");
}

int main(int argc, char **argv) {
    std::string scenePath, outPath, filmPath;
    LoadOptions opts;
    int spp = 0;
    std::vector<int> devices(1, 0);
    unsigned seed = 1;
    bool describeOnly = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&](const char *what) -> const char * {
            if (i + 1 >= argc) { fprintf(stderr, "%s needs a value\n", what); exit(usage()); }
            return argv[++i];
        };
        if (a == "-o") outPath = next("-o");
        else if (a == "--spp") spp = atoi(next("--spp"));
        else if (a == "--size") { opts.width = atoi(next("--size")); opts.height = atoi(next("--size")); }
        else if (a == "--seed") seed = (unsigned) strtoul(next("--seed"), nullptr, 10);
        else if (a == "--device") devices.assign(1, atoi(next("--device")));
        else if (a == "--devices") {
            devices.clear();
            const std::string list = next("--devices");
            for (size_t p = 0; p < list.size();) {
                const size_t q = list.find(',', p);
                const std::string tok = list.substr(p, q == std::string::npos ? std::string::npos : q - p);
                if (tok.empty() || tok.find_first_not_of("0123456789") != std::string::npos) { fprintf(stderr, "--devices expects a comma-separated list of device numbers\n"); return usage(); }
                devices.push_back(atoi(tok.c_str()));
                if (q == std::string::npos) break;
                p = q + 1;
            }
            if (devices.empty()) return usage();
        }
        else if (a == "--data-dir") opts.dataDir = next("--data-dir");
        else if (a == "--max-depth") opts.maxDepth = atoi(next("--max-depth"));
        else if (a == "--film") filmPath = next("--film");
        else if (a == "--describe") describeOnly = true;
        else if (a.rfind("-D", 0) == 0) {       // -Dname=value or -D name=value, like `mitsuba -D` (src/mitsuba/mitsuba.cpp:168-174)
            const std::string kv = a.size() > 2 ? a.substr(2) : std::string(next("-D"));
            const size_t eq = kv.find('=');
            if (eq == std::string::npos || eq == 0) { fprintf(stderr, "-D expects name=value\n"); return usage(); }
            opts.params[kv.substr(0, eq)] = kv.substr(eq + 1);
        }
        else if (!a.empty() && a[0] == '-') return usage();
        else scenePath = a;
    }
    if (scenePath.empty() || (!describeOnly && outPath.empty())) return usage();

    LoadedScene scene;
    const std::string err = loadSceneXml(scenePath, opts, scene);
    if (!err.empty()) { fprintf(stderr, "mtsamd-render: %s\n", err.c_str()); return 1; }
    if (!scene.integratorType.empty() && scene.integratorType != "path" && scene.integratorType != "direct" && scene.integratorType != "volpath_simple")
        fprintf(stderr, "mtsamd-render: the scene asks for integrator '%s'; rendering with 'path'\n", scene.integratorType.c_str());
    if (describeOnly) { describe(scene); return 0; }
    if (spp <= 0) spp = scene.sampleCount;

    if (mtsamd_abi_version() != MTSAMD_ABI_VERSION) {
        fprintf(stderr, "mtsamd-render: built against version %d of mtsamd.h, libmtsamd implements version %d\n", MTSAMD_ABI_VERSION, mtsamd_abi_version());
        return 1;
    }
    const mtsamd_scene_desc desc = scene.desc();
    std::vector<mtsamd_scene *> handles;
    for (int device : devices) {        // one replica of the scene per GPU
        mtsamd_scene *h = nullptr;
        if (mtsamd_init(device) != MTSAMD_OK || mtsamd_scene_create(&desc, &h) != MTSAMD_OK) { fprintf(stderr, "mtsamd-render: %s\n", mtsamd_last_error()); return 1; }
        handles.push_back(h);
    }
    mtsamd_scene *handle = handles[0];
    mtsamd_bvh_info info;
    mtsamd_scene_bvh_info(handle, &info);

    const int W = scene.film.crop_w, H = scene.film.crop_h;
    std::vector<float> film((size_t) W * H * 5, 0.0f), rgba((size_t) W * H * 4);
    mtsamd_render_opts ro;
    memset(&ro, 0, sizeof(ro));
    ro.spp = ro.spp_total = spp;
    ro.seed = seed;
    mtsamd_stats st;
    if (handles.size() == 1) {
        if (mtsamd_render(handle, &ro, film.data(), &st) != MTSAMD_OK) { fprintf(stderr, "mtsamd-render: %s\n", mtsamd_last_error()); return 1; }
    } else {
        std::vector<mtsamd_stats> per(handles.size());
        if (mtsamd_render_multi(handles.data(), (int32_t) handles.size(), &ro, 0, film.data(), per.data()) != MTSAMD_OK) { fprintf(stderr, "mtsamd-render: %s\n", mtsamd_last_error()); return 1; }
        memset(&st, 0, sizeof(st));
        for (const mtsamd_stats &p : per) {     // the job's ray counts; its duration is the slowest GPU's
            st.normal_rays += p.normal_rays; st.shadow_rays += p.shadow_rays; st.paths += p.paths;
            st.seconds = std::max(st.seconds, p.seconds);
        }
    }
    mtsamd_film_develop(handle, film.data(), rgba.data());
    // the extension picks the container: .png -> LDRFilm's tone-mapped 8-bit image (with the scene's ldrfilm parameters, or
    // its defaults), .exr -> OpenEXR, anything else -> PFM
    auto endsWith = [&](const char *ext) { return outPath.size() >= 4 && outPath.compare(outPath.size() - 4, 4, ext) == 0; };
    std::string werr;
    std::vector<Metadata> metadata;     // the film's annotations (PNG and OpenEXR carry them; PFM has no place for them)
    const std::string merr = resolveAnnotations(scene, outPath, st.seconds, (int) handles.size(), metadata);
    if (!merr.empty()) { fprintf(stderr, "mtsamd-render: %s\n", merr.c_str()); return 1; }
    if (endsWith(".png")) {
        const int channels[4] = { 1, 2, 3, 4 };
        std::vector<uint8_t> ldr((size_t) W * H * 4);
        if (mtsamd_film_develop_ldr(film.data(), W, H, &scene.ldr, ldr.data()) != MTSAMD_OK) { fprintf(stderr, "mtsamd-render: %s\n", mtsamd_last_error()); return 1; }
        werr = writePng(outPath, ldr.data(), W, H, channels[scene.ldr.pixel_format], &metadata);
    } else {
        werr = endsWith(".exr") ? writeExr(outPath, rgba.data(), W, H, scene.hdrPixelFormat, scene.hdrHalf, &metadata) : writePfm(outPath, rgba.data(), W, H);
    }
    for (mtsamd_scene *h : handles) mtsamd_scene_destroy(h);
    if (!werr.empty()) { fprintf(stderr, "mtsamd-render: %s\n", werr.c_str()); return 1; }
    if (!filmPath.empty()) {
        FILE *f = fopen(filmPath.c_str(), "wb");
        if (!f) { fprintf(stderr, "mtsamd-render: cannot write %s\n", filmPath.c_str()); return 1; }
        fwrite(film.data(), sizeof(float), film.size(), f);
        fclose(f);
    }
    const double rays = (double) st.normal_rays + (double) st.shadow_rays;
    printf("{\"width\": %d, \"height\": %d, \"spp\": %d, \"triangles\": %u, \"bvh_nodes\": %u, \"bvh_build_s\": %.3f, \"seconds\": %.4f, "
           "\"normal_rays\": %llu, \"shadow_rays\": %llu, \"mrays_per_s\": %.1f, \"mpix_spp_per_s\": %.1f, \"devices\": %zu}\n",
           W, H, spp, info.n_triangles, info.n_nodes, info.build_seconds, st.seconds, (unsigned long long) st.normal_rays,
           (unsigned long long) st.shadow_rays, rays / st.seconds / 1e6, (double) W * H * spp / st.seconds / 1e6, handles.size());
    return 0;
}
