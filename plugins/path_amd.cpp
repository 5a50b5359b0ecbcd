/*
 * path_amd -- Mitsuba 0.5 integrator plugin that runs the `path` integrator's render job on an MI355X through
 * libmtsamd.so. Drop this file into src/integrators/path/ of a Mitsuba 0.5 tree, add
 *     plugins += env.SharedLibrary('path_amd', ['path/path_amd.cpp'], LIBS=env['LIBS']+['mtsamd'])
 * to src/integrators/SConscript, and use <integrator type="path_amd"> with the parameters of `path`.
 *
 * It is NOT compiled in this repository (a Mitsuba build needs Boost/Xerces/OpenEXR); it documents, in compilable
 * form against the 0.5 headers, the binding a maintainer adds. Interfaces used on the Mitsuba side:
 *   Integrator::render / cancel                      include/mitsuba/render/integrator.h:70-84
 *   MonteCarloIntegrator members                     src/librender/integrator.cpp:190-225
 *   Scene::getMeshes/getShapes/getEmitters/getSensor include/mitsuba/render/scene.h
 *   TriMesh getters                                  include/mitsuba/render/trimesh.h:120-155
 *   ConfigurableObject::getProperties                include/mitsuba/core/cobject.h:77
 *   Film::setBitmap                                  include/mitsuba/render/film.h:55
 */
#include <mitsuba/render/scene.h>
#include <mitsuba/render/trimesh.h>
#include <mitsuba/render/renderqueue.h>
#include <mitsuba/core/fresolver.h>
#include <mitsuba/core/fstream.h>
#include <mitsuba/core/statistics.h>
#include <mtsamd.h>
#include <algorithm>
#include <map>
#include "../../bsdfs/ior.h"     /* lookupIOR: the named refractive indices of the BSDF plugins (src/bsdfs/ior.h:39-111) */

MTS_NAMESPACE_BEGIN

static StatsCounter amdNormalRays("MI355X path tracer", "Normal rays traced");
static StatsCounter amdShadowRays("MI355X path tracer", "Shadow rays traced");

class PathTracerAMD : public MonteCarloIntegrator {
public:
	PathTracerAMD(const Properties &props) : MonteCarloIntegrator(props) {
		m_device = props.getInteger("device", 0);
		/* "devices" = "0,1,2,3": render on several GPUs of the node (mtsamd_render_multi: the frame is dealt out as
		   interleaved full-width tiles, one host thread per GPU, films summed); default: the one GPU of "device" */
		m_deviceList = props.getString("devices", "");
		m_seed = (uint32_t) props.getInteger("seed", 1);
		m_refreshSamples = props.getInteger("refreshSamples", 0);
		/* "direct" = true: MIDirectIntegrator's estimator (src/integrators/direct/direct.cpp) with its sample counts */
		m_direct = props.getBoolean("direct", false);
		/* "simple" = true: the estimator of volpath_simple (src/integrators/path/volpath_simple.cpp) for scenes without media */
		m_simple = props.getBoolean("simple", false);
		if (m_direct && m_simple)
			Log(EError, "path_amd: 'direct' and 'simple' exclude each other");
		int shadingSamples = props.getInteger("shadingSamples", 1);
		m_emitterSamples = props.getInteger("emitterSamples", shadingSamples);
		m_bsdfSamples = props.getInteger("bsdfSamples", shadingSamples);
	}

	PathTracerAMD(Stream *stream, InstanceManager *manager) : MonteCarloIntegrator(stream, manager) {
		m_device = stream->readInt();
		m_deviceList = stream->readString();
		m_seed = stream->readUInt();
		m_refreshSamples = stream->readInt();
		m_direct = stream->readBool();
		m_simple = stream->readBool();
		m_emitterSamples = stream->readInt();
		m_bsdfSamples = stream->readInt();
	}

	void serialize(Stream *stream, InstanceManager *manager) const {
		MonteCarloIntegrator::serialize(stream, manager);
		stream->writeInt(m_device);
		stream->writeString(m_deviceList);
		stream->writeUInt(m_seed);
		stream->writeInt(m_refreshSamples);
		stream->writeBool(m_direct);
		stream->writeBool(m_simple);
		stream->writeInt(m_emitterSamples);
		stream->writeInt(m_bsdfSamples);
	}

	virtual ~PathTracerAMD() {
		for (size_t i = 0; i < m_handles.size(); ++i) mtsamd_scene_destroy(m_handles[i]);
	}

	/* The CPU path of SamplingIntegrator is not used; Li exists only to satisfy the interface. */
	Spectrum Li(const RayDifferential &, RadianceQueryRecord &) const {
		Log(EError, "path_amd renders whole jobs on the GPU; Li() is not available");
		return Spectrum(0.0f);
	}

	bool render(Scene *scene, RenderQueue *queue, const RenderJob *job, int sceneResID, int sensorResID, int samplerResID) {
		if (mtsamd_abi_version() != MTSAMD_ABI_VERSION)
			Log(EError, "path_amd was compiled against version %d of mtsamd.h, libmtsamd implements version %d",
				MTSAMD_ABI_VERSION, mtsamd_abi_version());
		if (m_handles.empty() && !upload(scene))
			return false;

		Film *film = scene->getSensor()->getFilm();
		const Vector2i size = film->getCropSize();
		const size_t spp = scene->getSampler()->getSampleCount();
		ref<Bitmap> accum = new Bitmap(Bitmap::EMultiChannel, Bitmap::EFloat32, size, 5);
		accum->clear();

		/* "refreshSamples" > 0: render that many samples per pixel at a time and hand the film over after each
		   chunk, so that mtsgui shows the image converging (RenderQueue::signalRefresh, src/librender/renderqueue.cpp;
		   the CPU integrators refresh per image block, RenderQueue::signalWorkEnd, renderqueue.cpp:119-123).
		   The chunks use disjoint sample indices of the same stream: the final film equals the one-call film. */
		const int32_t chunk = m_refreshSamples > 0 ? (int32_t) std::min<size_t>((size_t) m_refreshSamples, spp) : (int32_t) spp;
		ref<Bitmap> storage = new Bitmap(Bitmap::ESpectrumAlphaWeight, Bitmap::EFloat32, size);
		for (int32_t offset = 0; offset < (int32_t) spp; offset += chunk) {
			mtsamd_render_opts opts;
			memset(&opts, 0, sizeof(opts));
			opts.spp = std::min<int32_t>(chunk, (int32_t) spp - offset);
			opts.spp_total = (int32_t) spp;
			opts.sample_offset = offset;
			opts.seed = m_seed;
			std::vector<mtsamd_stats> stats(m_handles.size());
			int rv = m_handles.size() == 1 ? mtsamd_render(m_handles[0], &opts, accum->getFloat32Data(), &stats[0])
				: mtsamd_render_multi(&m_handles[0], (int32_t) m_handles.size(), &opts, 0, accum->getFloat32Data(), &stats[0]);
			if (rv == MTSAMD_ECANCELLED)
				return false;
			if (rv != MTSAMD_OK)
				Log(EError, "%s", mtsamd_last_error());
			for (size_t i = 0; i < stats.size(); ++i) {
				amdNormalRays += stats[i].normal_rays;
				amdShadowRays += stats[i].shadow_rays;
			}

			/* (R,G,B,alpha,weight) is HDRFilm's ESpectrumAlphaWeight storage when Spectrum is RGB
			   (src/films/hdrfilm.cpp:351-356); setBitmap stores it and develop() divides by the weight. */
			memcpy(storage->getFloat32Data(), accum->getFloat32Data(), (size_t) size.x * size.y * 5 * sizeof(float));
			film->setBitmap(storage);
			if (offset + chunk < (int32_t) spp)
				queue->signalRefresh(job);
		}
		queue->signalRefresh(job);
		return true;
	}

	void cancel() { for (size_t i = 0; i < m_handles.size(); ++i) mtsamd_cancel(m_handles[i]); }

	MTS_DECLARE_CLASS()
private:
	static void rgb(const Spectrum &s, float out[3]) { Float r, g, b; s.toLinearRGB(r, g, b); out[0] = r; out[1] = g; out[2] = b; }

	/* eta and k of `conductor` / `roughconductor` exactly as their constructors derive them (src/bsdfs/roughconductor.cpp:
	   174-190, src/bsdfs/conductor.cpp:155-176): the `material` preset (default "Cu") is read from data/ior/<name>.eta.spd
	   and .k.spd through the file resolver and converted to RGB unless it is "none"; explicit `eta` / `k` override the
	   preset; extEta may be a number or a named medium. The record takes the values BEFORE the division by extEta. */
	static void conductorConstants(const Properties &p, mtsamd_material &m) {
		ref<FileResolver> fResolver = Thread::getThread()->getFileResolver();
		std::string materialName = p.getString("material", "Cu");
		Spectrum intEta, intK;
		if (boost::to_lower_copy(materialName) == "none") {
			intEta = Spectrum(0.0f);
			intK = Spectrum(1.0f);
		} else {
			intEta.fromContinuousSpectrum(InterpolatedSpectrum(fResolver->resolve("data/ior/" + materialName + ".eta.spd")));
			intK.fromContinuousSpectrum(InterpolatedSpectrum(fResolver->resolve("data/ior/" + materialName + ".k.spd")));
		}
		rgb(p.getSpectrum("eta", intEta), m.eta);
		rgb(p.getSpectrum("k", intK), m.k);
		m.ext_eta = lookupIOR(p, "extEta", "air");
	}

	static int distribution(const Properties &p) {
		std::string d = boost::to_lower_copy(p.getString("distribution", "beckmann"));
		if (d == "beckmann") return MTSAMD_DISTR_BECKMANN;
		if (d == "ggx") return MTSAMD_DISTR_GGX;
		SLog(EError, "path_amd: microfacet distribution \"%s\" is not supported", d.c_str());
		return -1;
	}

	/* Constructor-level parameters come from the Properties each plugin was created with; the derived state
	   (energy-conservation scale, sampling weights, eta ratios) is recomputed by the library as configure() does. */
	bool material(const BSDF *bsdf, mtsamd_material &m, std::vector<std::vector<float> > &tables) {
		const Properties &p = bsdf->getProperties();
		const std::string type = p.getPluginName();
		memset(&m, 0, sizeof(m));
		m.sample_visible = p.getBoolean("sampleVisible", true) ? 1 : 0;
		m.ensure_energy_conservation = p.getBoolean("ensureEnergyConservation", true) ? 1 : 0;
		if (type == "diffuse") {
			m.type = MTSAMD_BSDF_DIFFUSE;
			rgb(p.getSpectrum(p.hasProperty("reflectance") ? "reflectance" : "diffuseReflectance", Spectrum(.5f)), m.diffuse_reflectance);
		} else if (type == "roughconductor") {
			m.type = MTSAMD_BSDF_ROUGHCONDUCTOR;
			m.distribution = distribution(p);
			Float a = p.getFloat("alpha", 0.1f);
			m.alpha_u = p.getFloat("alphaU", a); m.alpha_v = p.getFloat("alphaV", a);
			rgb(p.getSpectrum("specularReflectance", Spectrum(1.0f)), m.specular_reflectance);
			conductorConstants(p, m);
		} else if (type == "roughplastic") {
			m.type = MTSAMD_BSDF_ROUGHPLASTIC;
			m.distribution = distribution(p);
			m.alpha_u = m.alpha_v = p.getFloat("alpha", 0.1f);
			m.int_ior = lookupIOR(p, "intIOR", "polypropylene"); m.ext_ior = lookupIOR(p, "extIOR", "air");
			m.variant = p.getBoolean("nonlinear", false) ? 1 : 0;
			rgb(p.getSpectrum("diffuseReflectance", Spectrum(.5f)), m.diffuse_reflectance);
			rgb(p.getSpectrum("specularReflectance", Spectrum(1.0f)), m.specular_reflectance);
			/* RoughPlastic::configure loads data/microfacet/<distr>.dat through the file resolver (roughplastic.cpp:281-299) */
			fs::path path = Thread::getThread()->getFileResolver()->resolve(
				std::string("data/microfacet/") + (m.distribution == MTSAMD_DISTR_GGX ? "ggx" : "beckmann") + ".dat");
			ref<FileStream> fs = new FileStream(path, FileStream::EReadOnly);
			std::vector<uint8_t> dat(fs->getSize());
			fs->read(&dat[0], dat.size());
			tables.push_back(std::vector<float>(256));
			uint32_t n = 0;
			if (mtsamd_rtrans_reduce(&dat[0], dat.size(), m.int_ior / m.ext_ior, m.alpha_u, &tables.back()[0], 256, &n,
			                         &m.rtrans_diffuse_ext, &m.rtrans_diffuse_int) != MTSAMD_OK)
				Log(EError, "%s", mtsamd_last_error());
			m.rtrans = &tables.back()[0]; m.rtrans_n = n;
		} else if (type == "dielectric") {
			m.type = MTSAMD_BSDF_DIELECTRIC;
			m.int_ior = lookupIOR(p, "intIOR", "bk7"); m.ext_ior = lookupIOR(p, "extIOR", "air");
			rgb(p.getSpectrum("specularReflectance", Spectrum(1.0f)), m.specular_reflectance);
			rgb(p.getSpectrum("specularTransmittance", Spectrum(1.0f)), m.specular_transmittance);
		} else if (type == "ward") {
			m.type = MTSAMD_BSDF_WARD;
			std::string v = boost::to_lower_copy(p.getString("variant", "balanced"));
			m.variant = v == "ward" ? 0 : v == "ward-duer" ? 1 : 2;
			m.alpha_u = p.getFloat("alphaU", .1f); m.alpha_v = p.getFloat("alphaV", .1f);
			rgb(p.getSpectrum("diffuseReflectance", Spectrum(.5f)), m.diffuse_reflectance);
			rgb(p.getSpectrum("specularReflectance", Spectrum(.2f)), m.specular_reflectance);
		} else if (type == "roughdielectric") {
			m.type = MTSAMD_BSDF_ROUGHDIELECTRIC;
			m.distribution = distribution(p);
			Float a = p.getFloat("alpha", 0.1f);
			m.alpha_u = p.getFloat("alphaU", a); m.alpha_v = p.getFloat("alphaV", a);
			m.int_ior = lookupIOR(p, "intIOR", "bk7"); m.ext_ior = lookupIOR(p, "extIOR", "air");
			rgb(p.getSpectrum("specularReflectance", Spectrum(1.0f)), m.specular_reflectance);
			rgb(p.getSpectrum("specularTransmittance", Spectrum(1.0f)), m.specular_transmittance);
		} else if (type == "conductor") {
			m.type = MTSAMD_BSDF_CONDUCTOR;
			rgb(p.getSpectrum("specularReflectance", Spectrum(1.0f)), m.specular_reflectance);
			conductorConstants(p, m);
		} else if (type == "plastic") {
			m.type = MTSAMD_BSDF_PLASTIC;
			m.int_ior = lookupIOR(p, "intIOR", "polypropylene"); m.ext_ior = lookupIOR(p, "extIOR", "air");
			m.variant = p.getBoolean("nonlinear", false) ? 1 : 0;
			rgb(p.getSpectrum("diffuseReflectance", Spectrum(.5f)), m.diffuse_reflectance);
			rgb(p.getSpectrum("specularReflectance", Spectrum(1.0f)), m.specular_reflectance);
		} else {
			/* "twosided": the library takes it as two_sided / back_material on the nested BRDF's record, but
			   TwoSidedBRDF keeps its nested BRDFs private (src/bsdfs/twosided.cpp:231-232); with a getter added there,
			   fill the record from getNestedBRDF(0) and append the record of getNestedBRDF(1) for back_material.
			   "mixturebsdf" likewise: the library takes mix_count / mix_material / mix_weight, but MixtureBSDF keeps
			   m_bsdfs and m_weights private (src/bsdfs/mixturebsdf.cpp:317-322). */
			Log(EWarn, "path_amd: BSDF plugin \"%s\" is not on the GPU path", type.c_str());
			return false;
		}
		m.back_material = -1;
		return true;
	}

	bool upload(Scene *scene) {
		const std::vector<TriMesh *> &meshes = scene->getMeshes();
		if (meshes.size() != scene->getShapes().size()) {
			Log(EWarn, "path_amd: only triangle meshes are supported"); return false;
		}
		std::vector<float> pos, nrm, uv;
		std::vector<uint32_t> idx;
		std::vector<mtsamd_shape> shapes;
		std::vector<mtsamd_material> materials;
		std::vector<mtsamd_emitter> emitters;
		std::vector<std::vector<float> > tables;
		tables.reserve(meshes.size());
		std::map<const BSDF *, int> bsdfIds;
		std::map<const Emitter *, int32_t> emitterShape;
		bool anyN = false, anyUV = false;
		for (size_t i = 0; i < meshes.size(); ++i) { anyN |= meshes[i]->hasVertexNormals(); anyUV |= meshes[i]->hasVertexTexcoords(); }

		for (size_t i = 0; i < meshes.size(); ++i) {
			const TriMesh *mesh = meshes[i];
			const uint32_t base = (uint32_t) (pos.size() / 3);
			mtsamd_shape s;
			s.first_triangle = (uint32_t) (idx.size() / 3);
			s.n_triangles = (uint32_t) mesh->getTriangleCount();
			s.flags = (mesh->hasVertexNormals() ? MTSAMD_SHAPE_HAS_NORMALS : 0) | (mesh->hasVertexTexcoords() ? MTSAMD_SHAPE_HAS_TEXCOORDS : 0);
			for (size_t v = 0; v < mesh->getVertexCount(); ++v) {
				const Point &p = mesh->getVertexPositions()[v];
				pos.push_back(p.x); pos.push_back(p.y); pos.push_back(p.z);
				if (anyN) { Normal n = mesh->hasVertexNormals() ? mesh->getVertexNormals()[v] : Normal(0.0f); nrm.push_back(n.x); nrm.push_back(n.y); nrm.push_back(n.z); }
				if (anyUV) { Point2 t = mesh->hasVertexTexcoords() ? mesh->getVertexTexcoords()[v] : Point2(0.0f); uv.push_back(t.x); uv.push_back(t.y); }
			}
			for (size_t t = 0; t < mesh->getTriangleCount(); ++t)
				for (int k = 0; k < 3; ++k)
					idx.push_back(base + mesh->getTriangles()[t].idx[k]);

			const BSDF *bsdf = mesh->getBSDF();
			if (!bsdfIds.count(bsdf)) {
				mtsamd_material m;
				if (!material(bsdf, m, tables)) return false;
				bsdfIds[bsdf] = (int) materials.size();
				materials.push_back(m);
			}
			s.material = bsdfIds[bsdf];
			s.emitter = -1;
			if (mesh->isEmitter()) emitterShape[mesh->getEmitter()] = (int32_t) i;
			shapes.push_back(s);
		}
		/* emitter records in the order of Scene::getEmitters(), which is the order of the emitter choice
		   (src/librender/scene.cpp:375-380): area lights point at their mesh, the constant environment
		   (src/emitters/constant.cpp) has no shape, point / spot / directional lights carry their transform */
		const ref_vector<Emitter> &sceneEmitters = scene->getEmitters();
		for (size_t k = 0; k < sceneEmitters.size(); ++k) {
			const Emitter *e = sceneEmitters[k].get();
			const Properties &ep = e->getProperties();
			const std::string kind = ep.getPluginName();
			mtsamd_emitter me;
			memset(&me, 0, sizeof(me));
			me.shape = -1;
			me.sampling_weight = e->getSamplingWeight();
			const Matrix4x4 &M = e->getWorldTransform()->eval(0).getMatrix();
			for (int r = 0; r < 4; ++r)
				for (int c = 0; c < 4; ++c)
					me.to_world[4 * r + c] = (float) M(r, c);
			if (kind == "area") {
				std::map<const Emitter *, int32_t>::const_iterator it = emitterShape.find(e);
				if (it == emitterShape.end()) { Log(EWarn, "path_amd: area emitter without a triangle mesh"); return false; }
				me.shape = it->second;
				shapes[me.shape].emitter = (int32_t) emitters.size();
				rgb(ep.getSpectrum("radiance", Spectrum::getD65()), me.radiance);
			} else if (kind == "constant") {
				rgb(ep.getSpectrum("radiance", Spectrum::getD65()), me.radiance);
			} else if (kind == "point") {
				me.type = MTSAMD_EMITTER_POINT;
				rgb(ep.getSpectrum("intensity", Spectrum::getD65()), me.radiance);
			} else if (kind == "spot") {
				me.type = MTSAMD_EMITTER_SPOT;
				if (!ep.hasProperty("texture") || ep.getType("texture") == Properties::ESpectrum) {
					rgb(ep.getSpectrum("intensity", Spectrum(1.0f)) * ep.getSpectrum("texture", Spectrum::getD65()), me.radiance);
				} else { Log(EWarn, "path_amd: textured spot emitters are not on the GPU path"); return false; }
				me.cutoff_angle = (float) ep.getFloat("cutoffAngle", 20);
				me.beam_width = (float) ep.getFloat("beamWidth", me.cutoff_angle * 3.0f / 4.0f);
			} else if (kind == "directional") {
				me.type = MTSAMD_EMITTER_DIRECTIONAL;
				rgb(ep.getSpectrum("irradiance", Spectrum::getD65()), me.radiance);
			} else {
				Log(EWarn, "path_amd: emitter \"%s\" is not on the GPU path", kind.c_str());
				return false;
			}
			emitters.push_back(me);
		}

		mtsamd_scene_desc d;
		memset(&d, 0, sizeof(d));
		d.n_vertices = (uint32_t) (pos.size() / 3); d.n_triangles = (uint32_t) (idx.size() / 3);
		d.n_shapes = (uint32_t) shapes.size(); d.n_materials = (uint32_t) materials.size(); d.n_emitters = (uint32_t) emitters.size();
		d.positions = &pos[0]; d.normals = anyN ? &nrm[0] : NULL; d.texcoords = anyUV ? &uv[0] : NULL; d.indices = &idx[0];
		d.shapes = &shapes[0]; d.materials = &materials[0]; d.emitters = emitters.empty() ? NULL : &emitters[0];

		const PerspectiveCamera *cam = dynamic_cast<const PerspectiveCamera *>(scene->getSensor());
		if (!cam) { Log(EWarn, "path_amd: only the perspective sensor is supported"); return false; }
		const Matrix4x4 &m = cam->getWorldTransform()->eval(0).getMatrix();
		for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) d.camera.to_world[4 * r + c] = m(r, c);
		d.camera.x_fov = cam->getXFov(); d.camera.near_clip = cam->getNearClip(); d.camera.far_clip = cam->getFarClip();

		const Film *film = cam->getFilm();
		d.film.width = film->getSize().x; d.film.height = film->getSize().y;
		d.film.crop_x = film->getCropOffset().x; d.film.crop_y = film->getCropOffset().y;
		d.film.crop_w = film->getCropSize().x; d.film.crop_h = film->getCropSize().y;
		const ReconstructionFilter *rf = film->getReconstructionFilter();
		d.film.filter_radius = rf->getRadius();
		/* The filter's own table (ReconstructionFilter::configure, rfilter.cpp:37-55) read back entry by entry:
		   evalDiscretized truncates x * 31 / radius to an index (rfilter.h:76-77), so it is asked at the centre of
		   each cell, where rounding cannot land in the neighbouring entry. */
		for (int i = 0; i < 32; ++i)
			d.film.filter_values[i] = i < 31 ? rf->evalDiscretized(rf->getRadius() * (i + 0.5f) / 31.0f) : 0.0f;

		d.integrator.max_depth = m_maxDepth; d.integrator.rr_depth = m_rrDepth;
		d.integrator.strict_normals = m_strictNormals; d.integrator.hide_emitters = m_hideEmitters;
		d.integrator.type = m_direct ? MTSAMD_INTEGRATOR_DIRECT : m_simple ? MTSAMD_INTEGRATOR_VOLPATH_SIMPLE : MTSAMD_INTEGRATOR_PATH;
		d.integrator.emitter_samples = m_emitterSamples; d.integrator.bsdf_samples = m_bsdfSamples;

		/* one replica of the scene per GPU (Scene::initialize + kd-tree build happen once per worker node in the
		   reference's network rendering, src/libcore/sched_remote.cpp; here once per device) */
		std::vector<int> devices;
		if (m_deviceList.empty()) {
			devices.push_back(m_device);
		} else {
			std::vector<std::string> tokens = tokenize(m_deviceList, ", ");
			for (size_t i = 0; i < tokens.size(); ++i) devices.push_back(atoi(tokens[i].c_str()));
		}
		for (size_t i = 0; i < devices.size(); ++i) {
			mtsamd_scene *handle = NULL;
			if (mtsamd_init(devices[i]) != MTSAMD_OK || mtsamd_scene_create(&d, &handle) != MTSAMD_OK)
				Log(EError, "%s", mtsamd_last_error());
			m_handles.push_back(handle);
		}
		return true;
	}

	std::vector<mtsamd_scene *> m_handles;     /* one per GPU */
	int m_device;
	std::string m_deviceList;
	uint32_t m_seed;
	int m_refreshSamples;
	bool m_direct, m_simple;
	int m_emitterSamples, m_bsdfSamples;
};

MTS_IMPLEMENT_CLASS_S(PathTracerAMD, false, MonteCarloIntegrator)
MTS_EXPORT_PLUGIN(PathTracerAMD, "MI355X path tracer");
MTS_NAMESPACE_END
