// host_bench.cpp -- the host-pointer step (rbs_set_observation_f32 + rbs_loglikes) driven from
// C++ through the C-ABI, the way the reference's own (C++) filter would call it; bench.py's
// host_api_* leg goes through Python/ctypes and pays the interpreter per call.
//   host_bench [--prefetch | --borrowed | --loglikes-only] <workload.bin> <steps> [warmup]     (--loglikes-only: rbs_loglikes alone, the frame resident: SURVEY 8(d)'s metric to the letter)
//   (--prefetch: rbs_loglikes_prefetch + rbs_set_observation_prefetched, the next frame uploaded behind each call's kernels;
//    --borrowed: rbs_set_observation_borrowed_f32, the frame staged by rbs_loglikes behind its geometry kernel)
// workload.bin (written by bench.py, native endianness):
//   int32 rows, cols, n_objects, n, F, update; double K[9]; double params[7]
//   (p_occluded_visible, p_occluded_occluded, initial_occlusion_prob, tail_weight, model_sigma,
//   sigma_factor, delta_time); per object: int32 nv, nt; double v[3 nv]; int32 t[3 nt];
//   float frames[F][rows*cols]; double poses[F][n][n_objects][12]; int32 parents[n];
//   double tracker_init[n_objects][12] (position, rotation vector, velocities: the model-frame state the
//   device tracker starts from)
//   host_bench --plugin <workload.bin> <steps> [warmup]
// the same steps THROUGH THE PLUGIN SURFACE the reference drives: dbot_amd::RbSensorBuilder<State>(object_model,
// camera_data, parameters).build() (R:source/dbot_ros/tracker/particle_tracker_node.cpp:164-203), then per step
// sensor->set_observation(image) with the image as the reference hands it over -- a vector of rows*cols DOUBLES
// (R:source/dbot_ros/util/ros_interface.h:152-168, R:source/dbot_ros/object_tracker_ros.hpp:44-49) -- and
// sensor->loglikes(deltas, indices, update) with the particles' state DELTAS, one State (a heap vector of its own) per
// particle, around integrated_poses(); synchronous, the frame inside the clock, no look-ahead.  --plugin: the image is borrowed until
// loglikes returns (Options::borrow_frames, opt-in); --plugin-copy: copied at set_observation -- the default of the mirror and of the dbot binding.
//   host_bench --tracker-plugin <workload.bin> <particles>
// the device tracker THROUGH THE MIRROR of the reference's builders (dbot_amd::ParticleTrackerBuilder(...).build(), tracker->initialize,
// tracker->track(image of doubles) / submit + result): what dbot_ros's node would see.
//   host_bench --tracker <workload.bin> <particles>
// the device tracker (rbs_tracker_*: transition, loglikes, weights, KL test, resampling, mean; device
// RNG) over the workload's frames, frame by frame (rbs_tracker_track: one host synchronisation per
// frame, the frame uploaded from host memory) and with one frame of look-ahead (rbs_tracker_submit /
// rbs_tracker_result) -- what the reference's C++ node would see, no interpreter between the frames.
// Prints one line: "host_bench particle-likelihoods/s <v> ms/step <t> checksum <sum of finite log-likelihoods of the last step>".
#include <rbsensor_mi355x.h>
#include <dbot_amd/rb_sensor_builder.hpp>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

template <typename T>
static bool rd(std::FILE* f, T* p, size_t n) { return std::fread(p, sizeof(T), n, f) == n; }

int main(int argc, char** argv)
{
    const bool tracker_plugin = argc > 1 && !std::strcmp(argv[1], "--tracker-plugin");   // the tracker through dbot_amd::ParticleTrackerBuilder, images of doubles
    const bool tracker_mode = tracker_plugin || (argc > 1 && !std::strcmp(argv[1], "--tracker"));
    const bool prefetch_mode = argc > 1 && !std::strcmp(argv[1], "--prefetch");   // the next frame travels behind each call's kernels
    const bool loglikes_only = argc > 1 && !std::strcmp(argv[1], "--loglikes-only");   // SURVEY 8(d)'s metric to the letter: rbs_loglikes alone (poses up, log-likelihoods down), the frame resident
    const bool borrowed_mode = argc > 1 && !std::strcmp(argv[1], "--borrowed");   // rbs_set_observation_borrowed_f32: staged behind the geometry kernel
    const bool plugin_copy = argc > 1 && !std::strcmp(argv[1], "--plugin-copy");  // ... set_observation copying at once, as dbot's own sensors do
    const bool plugin_mode = plugin_copy || (argc > 1 && !std::strcmp(argv[1], "--plugin"));   // through dbot_amd::RbSensor (double frame, state deltas)
    if (tracker_mode || prefetch_mode || plugin_mode || borrowed_mode || loglikes_only) { --argc; ++argv; }
    if (argc < 3) { std::fprintf(stderr, "usage: host_bench workload.bin steps [warmup] | host_bench --tracker workload.bin particles\n"); return 2; }
    std::FILE* f = std::fopen(argv[1], "rb");
    if (!f) { std::perror(argv[1]); return 2; }
    const int steps = std::atoi(argv[2]), warmup = argc > 3 ? std::atoi(argv[3]) : 10;
    int32_t hd[6];
    double K[9], prm[7];
    if (!rd(f, hd, 6) || !rd(f, K, 9) || !rd(f, prm, 7)) return 2;
    const int rows = hd[0], cols = hd[1], nobj = hd[2], n = hd[3], F = hd[4], update = hd[5];
    std::vector<double> verts;
    std::vector<int32_t> tris, vcnt(nobj), tcnt(nobj);
    for (int b = 0; b < nobj; ++b) {
        int32_t c[2];
        if (!rd(f, c, 2)) return 2;
        vcnt[b] = c[0]; tcnt[b] = c[1];
        const size_t v0 = verts.size(), t0 = tris.size();
        verts.resize(v0 + 3 * (size_t)c[0]); tris.resize(t0 + 3 * (size_t)c[1]);
        if (!rd(f, verts.data() + v0, 3 * (size_t)c[0]) || !rd(f, tris.data() + t0, 3 * (size_t)c[1])) return 2;
    }
    const size_t npx = (size_t)rows * cols, stride = (size_t)12 * nobj * n;
    std::vector<float> frames(npx * F);
    std::vector<double> poses(stride * F), out(n);
    std::vector<int32_t> parents(n), idx(n);
    if (!rd(f, frames.data(), frames.size()) || !rd(f, poses.data(), poses.size()) || !rd(f, parents.data(), (size_t)n)) return 2;
    std::vector<double> init((size_t)12 * nobj, 0.0);
    const bool have_init = rd(f, init.data(), init.size());
    std::fclose(f);
    if (tracker_mode && !have_init) { std::fprintf(stderr, "host_bench --tracker: the workload holds no initial state\n"); return 2; }

    if (tracker_plugin) {
        // dbot_ros's own construction order (R:source/dbot_ros/tracker/particle_tracker_node.cpp:138-218): transition builder, sensor builder,
        // tracker builder -> build(); then tracker->initialize({state}) (:252) and, per frame, tracker->track(image) with the image a vector of
        // DOUBLES (R:source/dbot_ros/object_tracker_ros.hpp:44-49).  Frame by frame and with one frame of look-ahead (submit / result).
        using namespace dbot_amd;
        typedef FreeFloatingRigidBodiesState State;
        const int track_n = std::max(1, steps / nobj);
        std::vector<std::vector<Real>> vs(nobj);
        std::vector<std::vector<int32_t>> ts(nobj);
        size_t vo = 0, to = 0;
        for (int b = 0; b < nobj; ++b) {
            vs[b].assign(verts.begin() + vo, verts.begin() + vo + 3 * (size_t)vcnt[b]); vo += 3 * (size_t)vcnt[b];
            ts[b].assign(tris.begin() + to, tris.begin() + to + 3 * (size_t)tcnt[b]); to += 3 * (size_t)tcnt[b];
        }
        auto om = std::make_shared<ObjectModel>(vs, ts, false);
        auto cam = std::make_shared<CameraData>();
        for (int i = 0; i < 9; ++i) cam->camera_matrix[i] = K[i];
        cam->resolution.width = cols; cam->resolution.height = rows;
        RbSensorBuilder<State>::Parameters sp;
        sp.use_gpu = true; sp.sample_count = track_n;
        sp.occlusion.p_occluded_visible = prm[0]; sp.occlusion.p_occluded_occluded = prm[1]; sp.occlusion.initial_occlusion_prob = prm[2];
        sp.kinect.tail_weight = prm[3]; sp.kinect.model_sigma = prm[4]; sp.kinect.sigma_factor = prm[5]; sp.delta_time = prm[6];
        ObjectTransitionBuilder<State>::Parameters tpar;
        tpar.part_count = nobj;
        ParticleTrackerBuilder<ParticleTracker>::Parameters pp;
        pp.evaluation_count = track_n * nobj; pp.center_object_frame = false; pp.seed = 1;
        std::shared_ptr<ParticleTracker> tracker;
        try {
            tracker = ParticleTrackerBuilder<ParticleTracker>(std::make_shared<ObjectTransitionBuilder<State>>(tpar),
                                                              std::make_shared<RbSensorBuilder<State>>(om, cam, sp), om, pp).build();
        } catch (const std::exception& e) { std::printf("NO_DEVICE %s\n", e.what()); return 0; }
        std::vector<std::vector<double>> images(F, std::vector<double>(npx));
        for (int k = 0; k < F; ++k) for (size_t q = 0; q < npx; ++q) images[k][q] = (double)frames[npx * k + q];
        State st(nobj);
        st.data().assign(init.begin(), init.end());
        try {
            double dts[3], dts2[3];
            State est(nobj), est2(nobj);
            for (int rep = 0; rep < 3; ++rep) {
                tracker->initialize(std::vector<State>(1, st));
                tracker->track(images[0]);
                auto t0 = std::chrono::steady_clock::now();
                for (int k = 1; k < F; ++k) est = tracker->track(images[k]);
                dts[rep] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                tracker->initialize(std::vector<State>(1, st));
                tracker->track(images[0]);
                t0 = std::chrono::steady_clock::now();
                tracker->submit(images[1]);
                for (int k = 2; k < F; ++k) { tracker->submit(images[k]); est2 = tracker->result(); }
                est2 = tracker->result();
                dts2[rep] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            }
            std::sort(dts, dts + 3); std::sort(dts2, dts2 + 3);
            const bool same = est.data() == est2.data();
            std::printf("tracker_bench particles %d fps %.1f fps_pipelined %.1f resamplings %d identical %d state %.9g %.9g %.9g\n", track_n,
                        (F - 1) / dts[1], (F - 1) / dts2[1], tracker->resamplings(), same ? 1 : 0, est.data()[0], est.data()[1], est.data()[2]);
        } catch (const std::exception& e) { std::printf("ERROR %s\n", e.what()); return 1; }
        return 0;
    }

    if (plugin_mode) {
        using namespace dbot_amd;
        typedef FreeFloatingRigidBodiesState State;
        // construction in the node's order (particle_tracker_node.cpp:89-121, 164-203); the meshes are taken as they are
        // (center = false: the workload's poses are poses of these vertices)
        std::vector<std::vector<Real>> vs(nobj);
        std::vector<std::vector<int32_t>> ts(nobj);
        size_t vo = 0, to = 0;
        for (int b = 0; b < nobj; ++b) {
            vs[b].assign(verts.begin() + vo, verts.begin() + vo + 3 * (size_t)vcnt[b]); vo += 3 * (size_t)vcnt[b];
            ts[b].assign(tris.begin() + to, tris.begin() + to + 3 * (size_t)tcnt[b]); to += 3 * (size_t)tcnt[b];
        }
        auto om = std::make_shared<ObjectModel>(vs, ts, false);
        auto cam = std::make_shared<CameraData>();
        for (int i = 0; i < 9; ++i) cam->camera_matrix[i] = K[i];
        cam->resolution.width = cols; cam->resolution.height = rows;
        RbSensorBuilder<State>::Parameters p;
        p.use_gpu = true; p.sample_count = n;
        p.occlusion.p_occluded_visible = prm[0]; p.occlusion.p_occluded_occluded = prm[1]; p.occlusion.initial_occlusion_prob = prm[2];
        p.kinect.tail_weight = prm[3]; p.kinect.model_sigma = prm[4]; p.kinect.sigma_factor = prm[5]; p.delta_time = prm[6];
        std::shared_ptr<RbSensor<State>> sensor;
        try { sensor = RbSensorBuilder<State>(om, cam, p).build(); }
        catch (const std::exception& e) { std::printf("NO_DEVICE %s\n", e.what()); return 0; }
        sensor->borrow_observations(!plugin_copy);   // (--plugin: the opt-in borrowed frames; --plugin-copy: the default)
        // the images as the reference's tracker receives them: doubles
        std::vector<std::vector<double>> images(F, std::vector<double>(npx));
        for (int k = 0; k < F; ++k) for (size_t q = 0; q < npx; ++q) images[k][q] = (double)frames[npx * k + q];
        // deltas around the default pose = body b's pose of particle 0 in frame 0 (what a tracker's integrated_poses() would be
        // near); delta = pose (-) default:  R(delta) = R R0^T,  t(delta) = t - t0
        auto to_rotvec = [](const double* R, double* rv) {   // atan2 form (oracle/tracker_oracle.c matrix_to_rotvec, small angles)
            const double sx = 0.5 * (R[7] - R[5]), sy = 0.5 * (R[2] - R[6]), sz = 0.5 * (R[3] - R[1]);
            const double sn = std::sqrt(sx * sx + sy * sy + sz * sz), cs = 0.5 * ((R[0] + R[4] + R[8]) - 1.0);
            const double k = sn > 1e-12 ? std::atan2(sn, cs) / sn : 1.0;
            rv[0] = sx * k; rv[1] = sy * k; rv[2] = sz * k;
        };
        State& dflt = sensor->integrated_poses();
        for (int b = 0; b < nobj; ++b) {
            const double* P0 = poses.data() + 12 * (size_t)b;
            for (int k = 0; k < 3; ++k) dflt.position(b)[k] = P0[9 + k];
            to_rotvec(P0, dflt.euler_vector(b));
        }
        std::vector<std::vector<State>> deltas(F, std::vector<State>(n, State(nobj)));
        double R0[9];
        for (int k = 0; k < F; ++k)
            for (int i = 0; i < n; ++i)
                for (int b = 0; b < nobj; ++b) {
                    const double* Pp = poses.data() + stride * k + 12 * ((size_t)i * nobj + b);
                    State::rotation_matrix(dflt.euler_vector(b), R0);
                    double Rd[9];
                    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c)
                        Rd[3 * r + c] = Pp[3 * r] * R0[3 * c] + Pp[3 * r + 1] * R0[3 * c + 1] + Pp[3 * r + 2] * R0[3 * c + 2];   // R R0^T
                    State& d = deltas[k][i];
                    to_rotvec(Rd, d.euler_vector(b));
                    for (int q = 0; q < 3; ++q) d.position(b)[q] = Pp[9 + q] - dflt.position(b)[q];
                }
        RbSensor<State>::IntArray indices(n);
        RbSensor<State>::RealArray ll;
        auto pstep = [&](int i) {
            const int k = i % F;
            sensor->set_observation(images[k]);
            indices.assign(parents.begin(), parents.end());
            ll = sensor->loglikes(deltas[k], indices, update != 0);
        };
        try {
            for (int i = 0; i < warmup; ++i) pstep(i);
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < steps; ++i) pstep(warmup + i);
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            double sum = 0.0;
            for (double v : ll) if (std::isfinite(v)) sum += v;
            std::printf("host_bench particle-likelihoods/s %.1f ms/step %.5f checksum %.10g\n", (double)n * steps / dt, dt / steps * 1e3, sum);
        } catch (const std::exception& e) { std::printf("ERROR %s\n", e.what()); return 1; }
        return 0;
    }

    rbs_config cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.abi_version = RBS_ABI_VERSION;
    cfg.rows = rows; cfg.cols = cols;
    std::memcpy(cfg.K, K, sizeof K);
    const int track_n = tracker_mode ? std::max(1, steps / nobj) : 0;   // (--tracker: argv[2] = evaluations per frame)
    cfg.max_particles = tracker_mode ? track_n : n; cfg.n_objects = nobj;
    cfg.vertices = verts.data(); cfg.vertex_counts = vcnt.data();
    cfg.triangles = tris.data(); cfg.triangle_counts = tcnt.data();
    cfg.p_occluded_visible = prm[0]; cfg.p_occluded_occluded = prm[1]; cfg.initial_occlusion_prob = prm[2];
    cfg.tail_weight = prm[3]; cfg.model_sigma = prm[4]; cfg.sigma_factor = prm[5]; cfg.delta_time = prm[6];
    rbs_handle* h = nullptr;
    if (rbs_create(&cfg, &h) != RBS_OK) { std::printf("NO_DEVICE %s\n", h ? rbs_last_error(h) : "rbs_create failed"); return 0; }
    if (tracker_mode) {
        rbs_tracker_params tp;
        std::memset(&tp, 0, sizeof tp);
        for (int k = 0; k < 3; ++k) { tp.linear_sigma[k] = 0.0025; tp.angular_sigma[k] = 0.02; }   // R:config/particle_tracker.yaml:55-60
        tp.velocity_factor = 0.8; tp.max_kl_divergence = 2.0; tp.n_particles = track_n;
        rbs_tracker* t = nullptr;
        if (rbs_tracker_create(h, &tp, &t) != RBS_OK) { std::printf("ERROR %s\n", rbs_last_error(h)); return 1; }
        std::vector<double> state((size_t)12 * nobj), state2((size_t)12 * nobj);
        int32_t nres = 0;
        auto fail = [&]() { std::printf("ERROR %s\n", rbs_last_error(h)); return 1; };
        // the median of three passes over the sequence each way (a pass is 5 ms at 200 particles)
        double dts[3], dts2[3];
        for (int rep = 0; rep < 3; ++rep) {
            // frame by frame
            if (rbs_tracker_initialize(t, init.data())) return fail();
            if (rbs_tracker_track(t, frames.data(), nullptr, nullptr, 1, state.data(), &nres)) return fail();   // warm-up
            auto t0 = std::chrono::steady_clock::now();
            for (int k = 1; k < F; ++k)
                if (rbs_tracker_track(t, frames.data() + npx * k, nullptr, nullptr, 1, state.data(), &nres)) return fail();
            dts[rep] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            // one frame of look-ahead
            if (rbs_tracker_initialize(t, init.data())) return fail();
            if (rbs_tracker_track(t, frames.data(), nullptr, nullptr, 1, state2.data(), &nres)) return fail();
            t0 = std::chrono::steady_clock::now();
            if (rbs_tracker_submit(t, frames.data() + npx, nullptr, nullptr, 1)) return fail();
            for (int k = 2; k < F; ++k) {
                if (rbs_tracker_submit(t, frames.data() + npx * k, nullptr, nullptr, 1)) return fail();
                if (rbs_tracker_result(t, state2.data(), &nres)) return fail();
            }
            if (rbs_tracker_result(t, state2.data(), &nres)) return fail();
            dts2[rep] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
        std::sort(dts, dts + 3);
        std::sort(dts2, dts2 + 3);
        const double dt = dts[1], dt2 = dts2[1];
        const bool same = std::memcmp(state.data(), state2.data(), sizeof(double) * state.size()) == 0;
        std::printf("tracker_bench particles %d fps %.1f fps_pipelined %.1f resamplings %d identical %d state %.9g %.9g %.9g\n", track_n,
                    (F - 1) / dt, (F - 1) / dt2, (int)nres, same ? 1 : 0, state[0], state[1], state[2]);
        rbs_tracker_destroy(t);
        rbs_destroy(h);
        return 0;
    }
    auto step = [&](int i) -> int32_t {
        const int k = i % F;
        if (loglikes_only) { idx = parents; return rbs_loglikes(h, poses.data() + stride * k, idx.data(), n, update, out.data()); }
        if (int32_t rc = borrowed_mode ? rbs_set_observation_borrowed_f32(h, frames.data() + npx * k, npx) : rbs_set_observation_f32(h, frames.data() + npx * k, npx)) return rc;
        idx = parents;
        return rbs_loglikes(h, poses.data() + stride * k, idx.data(), n, update, out.data());
    };
    // --prefetch: rbs_loglikes_prefetch hands frame i + 1 over with step i's call, rbs_set_observation_prefetched makes it current
    auto step_ahead = [&](int i) -> int32_t {
        const int k = i % F, k1 = (i + 1) % F;
        idx = parents;
        if (int32_t rc = rbs_loglikes_prefetch(h, poses.data() + stride * k, idx.data(), n, update, out.data(), frames.data() + npx * k1, npx)) return rc;
        return rbs_set_observation_prefetched(h);
    };
    if ((prefetch_mode || loglikes_only) && rbs_set_observation_f32(h, frames.data(), npx)) { std::printf("ERROR %s\n", rbs_last_error(h)); return 1; }
    for (int i = 0; i < warmup; ++i)
        if (prefetch_mode ? step_ahead(i) : step(i)) { std::printf("ERROR %s\n", rbs_last_error(h)); return 1; }
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < steps; ++i)
        if (prefetch_mode ? step_ahead(warmup + i) : step(warmup + i)) { std::printf("ERROR %s\n", rbs_last_error(h)); return 1; }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    double sum = 0.0;
    for (double v : out) if (std::isfinite(v)) sum += v;
    std::printf("host_bench particle-likelihoods/s %.1f ms/step %.5f checksum %.10g\n", (double)n * steps / dt, dt / steps * 1e3, sum);
    rbs_destroy(h);
    return 0;
}
