// examples/prt_mp.cpp -- one process per GPU from a C++ host: every rank renders the tiles it owns, the library's
// RCCL gather (prt_hip_comm_* / prt_hip_gather_rccl) moves them to rank 0.
//
// The reference is one process with one Image (main.cpp:107-190): its tiles go to a thread pool.  Here the same 16x16 tiles
// (main.cpp:123-124) go to ranks (tile id % nranks); scene set-up uses the reference's classes (prt.h), the render and the
// exchange use the C-ABI (prt_hip.h), which is what a host in any language binds.
//
//   ./prt_mp [nranks [cornell|atrium [width height spp]]]     needs nranks MI355X (RCCL takes one rank per device)
//
// The parent forks the ranks BEFORE anything touches the GPU and hands rank 0's communicator id to the others through
// pipes.  Rank 0 then renders the whole image alone as well and requires the gathered image to equal it bit for bit.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <vector>

#include "../include/prt_hip.h"
#include "../prt_amd/csrc/host/prt.h"

using namespace prt;

static void check(int rc, const char* what, int rank)
{
    if (rc == PRT_HIP_OK) return;
    fprintf(stderr, "rank %d: %s failed (%d): %s\n", rank, what, rc, prt_hip_last_error());
    _exit(2);
}

static void setupScene(const char* which, Scene& scene, Camera& camera, uint32_t width, uint32_t height)
{
    scene.init();
    if (!strcmp(which, "atrium")) {
        Mesh m = SampleModels::getAtrium(262000, 1, true, true, 0.0f);
        m.calculateVertexNormals();
        auto b = new Bvh;
        b->build(std::move(m));
        scene.setDirectionalLight(normalize(Vector3f(0.05f, 1.0f, 0.1f)), Vector3f(16.7f, 15.6f, 11.7f)); // main.cpp:66
        scene.add(b);
        camera.create({-15.0f, 4.0f, 0.5f}, {1.0f, 0.08f, -0.05f}, width, height);
    } else {
        auto cbox = new Bvh;
        cbox->build(SampleModels::getCornellBox(true));
        scene.add(cbox);
        camera.create({0, 0.965, 2.6}, {0, 0, -1.0f}, width, height);
    }
}

static bool readAll(int fd, void* p, size_t n)
{
    char* c = (char*)p;
    while (n) {
        ssize_t r = read(fd, c, n);
        if (r <= 0) return false;
        c += r;
        n -= (size_t)r;
    }
    return true;
}

static int runRank(int rank, int nranks, int idPipeRead, const std::vector<int>& idPipeWrite, const char* which, uint32_t width,
                   uint32_t height, uint32_t spp)
{
    int devices = prt_hip_device_count();
    if (devices < nranks) {
        if (rank == 0) fprintf(stderr, "prt_mp: %d ranks need %d devices, this host shows %d\n", nranks, nranks, devices);
        return 3;
    }
    prt_hip_ctx* ctx = nullptr;
    check(prt_hip_create(rank, &ctx), "prt_hip_create", rank);

    // the communicator: rank 0 makes the id, the pipes carry its 128 bytes, every rank joins
    unsigned char id[PRT_HIP_COMM_ID_BYTES];
    if (rank == 0) {
        check(prt_hip_comm_unique_id(id), "prt_hip_comm_unique_id", rank);
        for (int fd : idPipeWrite)
            if (write(fd, id, sizeof id) != (ssize_t)sizeof id) { perror("write id"); return 2; }
    } else if (!readAll(idPipeRead, id, sizeof id)) {
        fprintf(stderr, "rank %d: no communicator id from rank 0\n", rank);
        return 2;
    }
    check(prt_hip_comm_init(ctx, id, rank, nranks), "prt_hip_comm_init", rank);

    // every rank holds the whole scene (pixels shard, the scene does not)
    Scene scene;
    Camera camera;
    setupScene(which, scene, camera, width, height);
    prt_scene_desc sd;
    Scene::DescStorage store;
    scene.describe(sd, store);
    prt_camera_desc cd;
    camera.describe(cd);
    check(prt_hip_upload_scene(ctx, &sd), "prt_hip_upload_scene", rank);
    check(prt_hip_set_camera(ctx, &cd), "prt_hip_set_camera", rank);

    prt_render_params p;
    memset(&p, 0, sizeof p);
    p.samples = spp;
    p.maxDepth = 14; // path_tracer.cpp:124
    p.rrDepth = 4;   // path_tracer.cpp:258
    p.seed = 12345;
    p.exposure = 1.0f;
    p.tileSize = 16; // main.cpp:123-124
    p.rank = (uint32_t)rank;
    p.nranks = (uint32_t)nranks;

    // two frames: the first pays RCCL's connection set-up, the second is the steady state
    double ms = 0.0;
    for (int frame = 0; frame < 2; frame++) {
        auto t0 = std::chrono::steady_clock::now();
        check(prt_hip_render(ctx, 0, 0, width - 1, height - 1, &p, nullptr, nullptr), "prt_hip_render", rank);
        check(prt_hip_gather_rccl(ctx, nullptr, 0, nullptr), "prt_hip_gather_rccl", rank);
        prt_hip_stats st;
        check(prt_hip_get_stats(ctx, &st), "prt_hip_get_stats", rank); // synchronises
        ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (frame == 1) {
            uint64_t bytes = 0;
            check(prt_hip_gather_payload_bytes(ctx, &bytes), "prt_hip_gather_payload_bytes", rank);
            printf("rank %d/%d: %.3f G rays, kernel %.2f ms, frame incl. gather %.2f ms, %.2f MB sent to the root\n", rank, nranks,
                   st.raysTraced * 1e-9, st.kernelMs, ms, rank == 0 ? 0.0 : bytes * 1e-6);
            fflush(stdout);
        }
    }

    int rc = 0;
    if (rank == 0) {
        Image gathered(width, height, true, 1.0f), alone(width, height, true, 1.0f);
        check(prt_hip_download(ctx, gathered.getPixels(), 0, 0, width - 1, height - 1), "prt_hip_download", rank);
        p.rank = 0;
        p.nranks = 1;
        check(prt_hip_render(ctx, 0, 0, width - 1, height - 1, &p, nullptr, nullptr), "prt_hip_render (alone)", rank);
        check(prt_hip_download(ctx, alone.getPixels(), 0, 0, width - 1, height - 1), "prt_hip_download (alone)", rank);
        bool same = memcmp(gathered.getPixels(), alone.getPixels(), sizeof(float) * 3 * width * height) == 0;
        printf("gathered image of %d ranks %s the one-GPU image\n", nranks, same ? "EQUALS" : "DIFFERS FROM");
        gathered.saveExr("prt_mp.exr");
        rc = same ? 0 : 1;
    }
    check(prt_hip_comm_destroy(ctx), "prt_hip_comm_destroy", rank);
    prt_hip_destroy(ctx);
    return rc;
}

int main(int argc, char** argv)
{
    int nranks = argc > 1 ? atoi(argv[1]) : 1;
    const char* which = argc > 2 ? argv[2] : "cornell";
    uint32_t width = argc > 3 ? (uint32_t)atoi(argv[3]) : 512, height = argc > 4 ? (uint32_t)atoi(argv[4]) : 512;
    uint32_t spp = argc > 5 ? (uint32_t)atoi(argv[5]) : 16;
    if (nranks < 1 || nranks > 8 || !width || !height || !spp || spp % 8) {
        fprintf(stderr, "usage: prt_mp [nranks 1..8 [cornell|atrium [width height spp(multiple of 8)]]]\n");
        return 2;
    }
    // pipes rank 0 -> rank r for the communicator id; made and forked before any GPU call in this process
    std::vector<int> rd(nranks, -1), wr;
    for (int r = 1; r < nranks; r++) {
        int fd[2];
        if (pipe(fd)) { perror("pipe"); return 2; }
        rd[r] = fd[0];
        wr.push_back(fd[1]);
    }
    std::vector<pid_t> pids;
    for (int r = 0; r < nranks; r++) {
        pid_t pid = fork();
        if (pid < 0) { perror("fork"); return 2; }
        if (pid == 0) {
            for (int q = 1; q < nranks; q++)
                if (q != r) close(rd[q]);
            if (r != 0)
                for (int fd : wr) close(fd);
            int rc = runRank(r, nranks, rd[r], wr, which, width, height, spp);
            fflush(stdout);
            _exit(rc);
        }
        pids.push_back(pid);
    }
    for (int fd : wr) close(fd);
    for (int r = 1; r < nranks; r++) close(rd[r]);
    int worst = 0;
    for (pid_t pid : pids) {
        int status = 0;
        waitpid(pid, &status, 0);
        int rc = WIFEXITED(status) ? WEXITSTATUS(status) : 128;
        if (rc > worst) worst = rc;
    }
    return worst;
}
