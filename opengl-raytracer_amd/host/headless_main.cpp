// Headless driver with the loop semantics of the reference's main.cpp:183-203, without a window:
//   * camera = yRotation(yaw) * xRotation(pitch) (main.cpp:41-51), pitch clamped to +-pi/2, WASD moves along the
//     camera axes at CAMERA_MOVE_SPEED units per second (main.cpp:216-231) -- driven by a script instead of GLFW;
//   * any input resets frameCount to 0 (main.cpp:184-188), otherwise frames accumulate through two ping-pong
//     frame buffers (main.cpp:189-201);
//   * "Raytracer initialised in ... seconds" and "Average FPS" are printed as the reference prints them.
// Display, where wanted, is a plain download of the frame; here the last frame is written as PFM/PPM (the image
// is stored bottom row first, as GL textures are; PFM keeps that order, PPM is flipped).
//
//   * "mode:N" in the script is the keys 1-4 (main.cpp:232-247): switch the render mode and restart the accumulation.
//
// Multi-GPU split for C++ hosts (one process per GPU, no counterpart in the reference): --rank r --world n renders this
// rank's share of every accumulation and writes it as a partial image; tools/combine_partials.py (or one ncclReduce /
// gather in a host that links RCCL) puts the shares together.
//   --partition samples  rank r renders the frames f of each accumulation with f mod n == r and writes their SUM
//   --partition tiles    rank r renders its interleaved band (stripes of 16 rows) of every frame, zeros elsewhere
//
// usage: rt_headless <scene.obj> [--size W H] [--frames N] [--pos x y z] [--pitch p] [--yaw y] [--mode 1..4]
//                    [--script "w:0.5,a:0.2,yaw:0.1,pitch:-0.2,mode:3,wait:8;..."] [--batch] [--out image.pfm|.ppm]
//                    [--device d] [--rank r --world n [--partition samples|tiles]] [--rccl <id file>]
//
// --rccl <id file>: the ranks (one process per GPU, started by any launcher) combine their shares themselves with one RCCL
// ncclReduce of the device frame buffers to rank 0, which scales the sum and writes --out; the 128-byte ncclUniqueId
// travels through the named file (rank 0 writes it, the others wait for it).  librccl is loaded only in this mode.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "raytrace.h"
#include "rt_host.h"

struct Step {
    float w = 0, a = 0, s = 0, d = 0;  // seconds each key is held
    double dyaw = 0, dpitch = 0;       // mouse look (radians)
    int wait = 1;                      // frames rendered after this input
    int mode = 0;                      // 1..4: render-mode key pressed (0: none)
};

static std::vector<Step> parse_script(const std::string& text) {
    std::vector<Step> out;
    std::stringstream all(text);
    std::string item;
    while (std::getline(all, item, ';')) {
        Step st;
        std::stringstream one(item);
        std::string kv;
        while (std::getline(one, kv, ',')) {
            auto c = kv.find(':');
            if (c == std::string::npos) continue;
            std::string k = kv.substr(0, c);
            double v = std::atof(kv.substr(c + 1).c_str());
            if (k == "w") st.w = (float)v;
            else if (k == "a") st.a = (float)v;
            else if (k == "s") st.s = (float)v;
            else if (k == "d") st.d = (float)v;
            else if (k == "yaw") st.dyaw = v;
            else if (k == "pitch") st.dpitch = v;
            else if (k == "wait") st.wait = std::max(1, (int)v);
            else if (k == "mode") st.mode = (int)v;
        }
        out.push_back(st);
    }
    return out;
}

static Vec3 mat_mul(const Mat3& m, Vec3 v) {  // column-major mat3 * vec3
    return {m[0] * v[0] + m[3] * v[1] + m[6] * v[2], m[1] * v[0] + m[4] * v[1] + m[7] * v[2],
            m[2] * v[0] + m[5] * v[1] + m[8] * v[2]};
}

static void write_image(const std::string& path, const std::vector<float>& rgba, int w, int h) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write " + path);
    bool pfm = path.size() > 4 && path.substr(path.size() - 4) == ".pfm";
    if (pfm) {
        std::fprintf(f, "PF\n%d %d\n-1.0\n", w, h);  // little endian, bottom row first (as stored)
        for (int i = 0; i < w * h; i++) std::fwrite(&rgba[4 * (size_t)i], sizeof(float), 3, f);
    } else {
        std::fprintf(f, "P6\n%d %d\n255\n", w, h);
        for (int y = h - 1; y >= 0; y--)
            for (int x = 0; x < w; x++)
                for (int c = 0; c < 3; c++) {
                    float v = std::min(1.0f, std::max(0.0f, rgba[4 * ((size_t)y * w + x) + c]));
                    std::fputc((int)(v * 255.0f + 0.5f), f);
                }
    }
    std::fclose(f);
}

// RCCL, loaded on demand (the library is half a gigabyte; the single-GPU driver should not pay for mapping it).
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclComm_t comm = nullptr;

    void check(ncclResult_t r, const char* what) const {
        if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + (GetErrorString ? GetErrorString(r) : "RCCL error"));
    }
    void load() {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
            if ((lib = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
        if (!lib) throw std::runtime_error(std::string("cannot load librccl: ") + dlerror());
        auto sym = [&](const char* n) {
            void* p = dlsym(lib, n);
            if (!p) throw std::runtime_error(std::string("librccl lacks ") + n);
            return p;
        };
        GetUniqueId = (decltype(GetUniqueId))sym("ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))sym("ncclCommInitRank");
        Reduce = (decltype(Reduce))sym("ncclReduce");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
    }
    // rank 0 creates the id and publishes it through the file (written under another name, then renamed, so that a
    // reader never sees half of it); the other ranks wait for the file.  A file left behind by an earlier run must not be
    // mistaken for this run's: rank 0 removes any before it publishes and removes its own once ncclCommInitRank has
    // returned (every rank has read it by then: the call completes only when all ranks have joined), and the other
    // ranks ignore a file written more than a minute before they started (all ranks of a run are on one node and are
    // started together).  Communicators of more than one rank have not run on hardware yet: the build box has one GPU.
    void init(const std::string& id_file, int rank, int world) {
        const time_t started = std::time(nullptr);
        load();
        ncclUniqueId id;
        if (rank == 0) {
            (void)std::remove(id_file.c_str());
            check(GetUniqueId(&id), "ncclGetUniqueId");
            const std::string tmp = id_file + ".tmp";
            FILE* f = std::fopen(tmp.c_str(), "wb");
            if (!f || std::fwrite(&id, sizeof id, 1, f) != 1) throw std::runtime_error("cannot write " + tmp);
            std::fclose(f);
            if (std::rename(tmp.c_str(), id_file.c_str()) != 0) throw std::runtime_error("cannot publish " + id_file);
        } else {
            bool ok = false;
            for (int tries = 0; tries < 1200 && !ok; tries++) {  // up to two minutes
                struct stat sb;
                if (stat(id_file.c_str(), &sb) == 0 && sb.st_mtime + 60 >= started) {
                    FILE* f = std::fopen(id_file.c_str(), "rb");
                    if (f) {
                        ok = std::fread(&id, sizeof id, 1, f) == 1;
                        std::fclose(f);
                    }
                }
                if (!ok) usleep(100000);
            }
            if (!ok) throw std::runtime_error("no ncclUniqueId of this run appeared in " + id_file);
        }
        check(CommInitRank(&comm, world, id, rank), "ncclCommInitRank");
        if (rank == 0) (void)std::remove(id_file.c_str());
    }
    void shutdown() {
        if (comm) (void)CommDestroy(comm);
        comm = nullptr;
    }
};

#define HIP_OK(call)                                                                                        \
    do {                                                                                                    \
        hipError_t e_ = (call);                                                                             \
        if (e_ != hipSuccess) throw std::runtime_error(std::string(#call) + ": " + hipGetErrorString(e_));  \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s <scene.obj> [--size W H] [--frames N] [--pos x y z] [--pitch p] [--yaw y] "
                             "[--mode m] [--script s] [--batch] [--out file]\n", argv[0]);
        return 1;
    }
    std::string obj = argv[1], out_path, script, rccl_id_file;
    int W = 1920, H = 1080, frames = 64, mode = RT_MODE_RENDER;
    bool batch = false, tiles = false;
    int rank = 0, world = 1, device = 0;
    Vec3 cameraPos = CAMERA_START_POS;
    double cameraPitch = 0.0, cameraYaw = 0.0;
    for (int i = 2; i < argc; i++) {
        std::string a = argv[i];
        auto need = [&](int n) { if (i + n >= argc) throw std::runtime_error("missing value after " + a); };
        if (a == "--size") { need(2); W = std::atoi(argv[++i]); H = std::atoi(argv[++i]); }
        else if (a == "--frames") { need(1); frames = std::atoi(argv[++i]); }
        else if (a == "--pos") { need(3); for (int k = 0; k < 3; k++) cameraPos[k] = (float)std::atof(argv[++i]); }
        else if (a == "--pitch") { need(1); cameraPitch = std::atof(argv[++i]); }
        else if (a == "--yaw") { need(1); cameraYaw = std::atof(argv[++i]); }
        else if (a == "--mode") { need(1); mode = std::atoi(argv[++i]); }
        else if (a == "--script") { need(1); script = argv[++i]; }
        else if (a == "--out") { need(1); out_path = argv[++i]; }
        else if (a == "--batch") batch = true;
        else if (a == "--rank") { need(1); rank = std::atoi(argv[++i]); }
        else if (a == "--world") { need(1); world = std::atoi(argv[++i]); }
        else if (a == "--device") { need(1); device = std::atoi(argv[++i]); }
        else if (a == "--partition") { need(1); tiles = std::string(argv[++i]) == "tiles"; }
        else if (a == "--rccl") { need(1); rccl_id_file = argv[++i]; }
    }
    if (world < 1 || rank < 0 || rank >= world) {
        std::fprintf(stderr, "error: --rank %d out of range for --world %d\n", rank, world);
        return 1;
    }
    const bool use_rccl = !rccl_id_file.empty();
    const bool partitioned = world > 1 || use_rccl;  // with --rccl also a single rank goes through the sum + reduce + scale path
    try {
        Rccl rccl;
        if (use_rccl) {
            HIP_OK(hipSetDevice(device));
            rccl.init(rccl_id_file, rank, world);
        }
        SceneData* sceneData = loadScene(obj);
        auto t0 = std::chrono::steady_clock::now();
        Raytracer* rt = raytraceInit(sceneData, W, H, device);
        std::cout << "Raytracer initialised in " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()
                  << " seconds" << std::endl;
        size_t bytes = (size_t)W * H * 4 * sizeof(float);
        float *currentFrame = nullptr, *prevFrame = nullptr;
        HIP_OK(hipMalloc(&currentFrame, bytes));
        HIP_OK(hipMalloc(&prevFrame, bytes));
        HIP_OK(hipMemset(currentFrame, 0, bytes));
        HIP_OK(hipMemset(prevFrame, 0, bytes));

        std::vector<Step> steps = parse_script(script);
        if (steps.empty()) steps.push_back(Step{0, 0, 0, 0, 0, 0, frames});
        int frameCount = 0, totalFrames = 0;
        Mat3 cameraRotation{};
        std::cout << "BEGAN RENDER LOOP" << std::endl;
        auto start = std::chrono::steady_clock::now();
        for (size_t si = 0; si < steps.size(); si++) {
            const Step& st = steps[si];
            bool clearAccumulatedFrames = si > 0 || st.w || st.a || st.s || st.d || st.dyaw != 0 || st.dpitch != 0;
            if (st.mode >= RT_MODE_RENDER && st.mode <= RT_MODE_REFLECTIONS) {  // keys 1-4, main.cpp:232-247
                mode = st.mode;
                clearAccumulatedFrames = true;
            }
            // processInput: translate along the camera axes with the rotation of the previous frame
            cameraPos = {cameraPos[0], cameraPos[1], cameraPos[2]};
            Vec3 mv = mat_mul(cameraRotation, {CAMERA_MOVE_SPEED * (st.d - st.a), 0.0f, CAMERA_MOVE_SPEED * (st.w - st.s)});
            if (st.w || st.a || st.s || st.d) for (int k = 0; k < 3; k++) cameraPos[k] += mv[k];
            // updateCameraRotation
            cameraPitch = std::max(-M_PI / 2.0, std::min(M_PI / 2.0, cameraPitch + st.dpitch));
            cameraYaw += st.dyaw;
            rt_host_camera_rotation(cameraPitch, cameraYaw, cameraRotation.data());
            if (clearAccumulatedFrames) frameCount = 0;
            if (partitioned) {
                if (mode != RT_MODE_RENDER) throw std::runtime_error("--world > 1 supports render mode 1 only");
                if (frameCount == 0) HIP_OK(hipMemset(currentFrame, 0, bytes));  // a fresh accumulation: sums and bands start from zero
                if (tiles) {
                    raytraceAccumulatePart(rt, cameraPos, cameraRotation, frameCount, 1, st.wait, false, rank, world, currentFrame);
                } else {
                    // the frames f of [frameCount, frameCount + wait) with f mod world == rank
                    const int end = frameCount + st.wait;
                    const int first = frameCount + ((rank - frameCount) % world + world) % world;
                    if (first < end)
                        raytraceAccumulatePart(rt, cameraPos, cameraRotation, first, world, (end - first + world - 1) / world, true, 0, 0,
                                               currentFrame);
                }
                frameCount += st.wait;
                totalFrames += st.wait;
            } else if (batch && mode == RT_MODE_RENDER) {
                // same pixels, but the library may render the frames of one call concurrently
                raytraceAccumulate(rt, cameraPos, cameraRotation, frameCount, st.wait, currentFrame);
                frameCount += st.wait;
                totalFrames += st.wait;
            } else {
                for (int f = 0; f < st.wait; f++) {
                    raytrace(rt, cameraPos, cameraRotation, mode, frameCount, prevFrame, currentFrame);
                    std::swap(currentFrame, prevFrame);
                    frameCount++;
                    totalFrames++;
                }
                std::swap(currentFrame, prevFrame);  // currentFrame = the frame written last
            }
        }
        if (use_rccl) {
            // the one collective of the path: sum (samples) or gather (tiles: disjoint bands, zero elsewhere) to rank 0,
            // on the stream the render was issued on
            void* st = nullptr;
            if (rt_get_stream(rt->ctx, &st) != RT_OK) throw std::runtime_error(rt_last_error(rt->ctx));
            rccl.check(rccl.Reduce(currentFrame, currentFrame, (size_t)W * H * 4, ncclFloat, ncclSum, 0, rccl.comm, (hipStream_t)st), "ncclReduce");
        }
        rt_synchronize(rt->ctx);
        double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
        std::cout << "Average FPS: " << totalFrames / secs << "\n";
        if (!out_path.empty() && (!use_rccl || rank == 0)) {
            std::vector<float> host((size_t)W * H * 4);
            HIP_OK(hipMemcpy(host.data(), currentFrame, bytes, hipMemcpyDeviceToHost));
            if (use_rccl && !tiles) {  // sum of the frames of the last accumulation -> mean
                const float scale = 1.0f / (float)frameCount;
                for (float& v : host) v *= scale;
            }
            write_image(out_path, host, W, H);
            std::cout << "wrote " << out_path << std::endl;
        }
        (void)hipFree(currentFrame);
        (void)hipFree(prevFrame);
        rccl.shutdown();
        raytraceShutdown(rt);
        delete sceneData;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 2;
    }
    return 0;
}
