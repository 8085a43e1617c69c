// llama2_main.cpp -- the reference's command line (src/main.zig:800-1051) over the
// MI355X forward pass.  Same flags, defaults, clamping, output and tokens/s rule;
// transformer() runs on the GPU through include/llama2_hip.h.
//
//   llama2 <checkpoint> [-t temp] [-p top_p] [-n steps] [-i prompt] [-s seed] [-v] [-z tokenizer]
//          [-g n_gpus]   (extension: rows / heads sharded over n GPUs, one process per GPU)
//
// At -t 0 the whole generation loop runs on the device (l2z_greedy_run) and the host
// only prints; otherwise one l2z_transformer + l2z_logits_read per position feeds the
// reference's host-side samplers.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/llama2_hip.h"
#include "llama2_host.hpp"

using namespace l2zhost;

// main.zig:800-813
static const char *usage_text =
    "Usage:   llama2 <checkpoint> [options]\n"
    "Example: llama2 checkpoint.bin -n 256 -i \"Once upon a time\"\n"
    "Options:\n"
    " -h, --help                print this help message\n"
    " -t, --temperature <float> temperature, default 1.0 (0.0, 1]\n"
    " -p, --top-p <float>       p value in top-p (nucleus) sampling. default 0.9, 0 || 1 = off\n"
    " -n, --seq-len <int>       number of steps to run for, default 256. 0 = max_seq_len\n"
    " -i, --input <string>      input text for the prompt, default \"\"\n"
    " -s, --seed <int>          random seed, default to time\n"
    " -v, --verbose             print model info and tokens/s\n"
    " -z, --tokenizer <path>    path to the tokenizer to use, default to \"tokenizer.bin\"\n"
    " --tokens                  (extension) also print the token ids to stderr, one line\n"
    " -g, --gpus <int>          (extension) shard weight rows / heads over this many GPUs, default 1\n";

static bool verbose = false;
#define LOGV(...)                                 \
    do {                                          \
        if (verbose) fprintf(stderr, __VA_ARGS__); \
    } while (0)

static int die(const char *what)
{
    fprintf(stderr, "error: %s: %s\n", what, l2z_last_error());
    return 1;
}

// ---- -g N: one process per GPU.  The parent is rank 0 and forks ranks 1..N-1 BEFORE
// anything touches the GPU; every rank maps the checkpoint itself and uploads only its own rows
// (l2z_weights_init with a comm), the ranks exchange the 64-byte IPC handles of their landing
// arenas through files in a private temporary directory, and then all of them run the very same
// generation loop -- same tokens in, bit-identical logits out, so they stay in step
// without talking to each other; only rank 0 prints.
static int g_rank = 0, g_world = 1;

// A rank that cannot take part (comm_init / p2p_export failed on its GPU) leaves an error marker so that
// the others stop waiting at once instead of polling for its handle for a minute.
static void mark_failed(const std::string &dir)
{
    if (dir.empty()) return;
    FILE *f = fopen((dir + "/e" + std::to_string(g_rank)).c_str(), "wb");
    if (f) fclose(f);
}

static bool any_rank_failed(const std::string &dir, const std::vector<pid_t> &kids)
{
    for (int r = 0; r < g_world; r++)
        if (access((dir + "/e" + std::to_string(r)).c_str(), F_OK) == 0) return true;
    for (pid_t k : kids) {  // rank 0 only: a child that has already gone will never write its handle
        siginfo_t si;
        si.si_pid = 0;
        if (waitid(P_PID, (id_t)k, &si, WEXITED | WNOHANG | WNOWAIT) == 0 && si.si_pid == k) return true;
    }
    return false;
}

static bool exchange_handles(const std::string &dir, const std::vector<pid_t> &kids, const void *mine,
                             std::vector<char> *all)
{
    const std::string tmp = dir + "/h" + std::to_string(g_rank) + ".tmp", fin = dir + "/h" + std::to_string(g_rank);
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return false;
    const bool wrote = fwrite(mine, 1, L2Z_COMM_IPC_BYTES, f) == L2Z_COMM_IPC_BYTES;
    if (fclose(f) != 0 || !wrote) return false;
    if (rename(tmp.c_str(), fin.c_str()) != 0) return false;
    all->resize((size_t)g_world * L2Z_COMM_IPC_BYTES);
    for (int r = 0; r < g_world; r++) {
        const std::string p = dir + "/h" + std::to_string(r);
        FILE *g = nullptr;
        for (int tries = 0; tries < 6000 && !(g = fopen(p.c_str(), "rb")); tries++) {
            if ((tries % 10) == 9 && any_rank_failed(dir, kids)) return false;
            usleep(10000);
        }
        if (!g) return false;
        const size_t n = fread(all->data() + (size_t)r * L2Z_COMM_IPC_BYTES, 1, L2Z_COMM_IPC_BYTES, g);
        fclose(g);
        if (n != L2Z_COMM_IPC_BYTES) return false;
    }
    return true;
}

int main(int argc, char **argv)
{
    if (argc < 2) {  // :833-836
        fputs(usage_text, stdout);
        return 0;
    }
    const char *bin_path = nullptr;
    const char *input = nullptr;
    float temperature = 1.0f, top_p = 0.9f;  // :840-841
    size_t seq_len = 0;
    std::string tokenizer_path = "tokenizer.bin";
    bool dump_tokens = false;
    int n_gpus = 1;
    Prng prng((uint64_t)std::chrono::system_clock::now().time_since_epoch().count());  // :844-845

    auto need = [&](int &i, const char *what) -> const char * {  // :863-867 etc.
        if (++i >= argc) {
            fprintf(stderr, "error: missing argument for %s\n", what);
            exit(1);
        }
        return argv[i];
    };
    for (int i = 1; i < argc; i++) {  // :848-934
        const std::string a = argv[i];
        if (a == "-h" || a == "--help") {
            fputs(usage_text, stdout);
            return 0;
        }
        if (a.empty() || a[0] != '-') {
            if (bin_path) {  // :856-858
                fprintf(stderr, "error: multiple checkpoint paths specified\n");
                return 1;
            }
            bin_path = argv[i];
        } else if (a == "-t" || a == "--temperature") {
            const char *v = need(i, "temperature");
            char *end = nullptr;
            temperature = strtof(v, &end);
            if (end == v || *end) {
                fprintf(stderr, "unable to parse --temperature argument '%s'\n", v);
                return 1;
            }  // not clamped, :874
        } else if (a == "-n" || a == "--seq-len") {
            const char *v = need(i, "seq-len");
            char *end = nullptr;
            const long long n = strtoll(v, &end, 10);
            if (end == v || *end || n < 0) {
                fprintf(stderr, "unable to parse --seq-len argument '%s'\n", v);
                return 1;
            }
            seq_len = (size_t)n;
        } else if (a == "-p" || a == "--top-p") {
            const char *v = need(i, "top-p");
            char *end = nullptr;
            top_p = strtof(v, &end);
            if (end == v || *end) {
                fprintf(stderr, "unable to parse --top-p argument '%s'\n", v);
                return 1;
            }
            top_p = top_p < 0.0f ? 0.0f : (top_p > 1.0f ? 1.0f : top_p);  // :899
        } else if (a == "-i" || a == "--input") {
            input = need(i, "input");
        } else if (a == "-z" || a == "--tokenizer") {
            tokenizer_path = need(i, "tokenizer");
        } else if (a == "-s" || a == "--seed") {
            const char *v = need(i, "seed");
            char *end = nullptr;
            const unsigned long long s = strtoull(v, &end, 10);
            if (end == v || *end) {
                fprintf(stderr, "unable to parse --seed argument '%s'\n", v);
                return 1;
            }
            prng.seed_with((uint64_t)s);  // :926
        } else if (a == "-v" || a == "--verbose") {
            verbose = true;
        } else if (a == "--tokens") {
            dump_tokens = true;
        } else if (a == "-g" || a == "--gpus") {
            n_gpus = atoi(need(i, "gpus"));
            if (n_gpus < 1 || n_gpus > 16) {
                fprintf(stderr, "unable to use --gpus argument '%s'\n", argv[i]);
                return 1;
            }
        } else {  // :929-933
            fprintf(stderr, "error: unknown argument '%s'\n", argv[i]);
            fputs(usage_text, stdout);
            return 0;
        }
    }
    if (!bin_path) {
        fputs(usage_text, stdout);
        return 1;
    }

    // ---- ranks (before any GPU call: the children must start with a clean HIP state)
    std::string xdir;
    std::vector<pid_t> kids;
    if (n_gpus > 1) {
        setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);  // dmabuf IPC
        char tmpl[] = "/tmp/llama2_ranks_XXXXXX";
        if (!mkdtemp(tmpl)) {
            fprintf(stderr, "error: cannot create a temporary directory for the rank hand-shake\n");
            return 1;
        }
        xdir = tmpl;
        g_world = n_gpus;
        fflush(stdout);
        fflush(stderr);
        for (int r = 1; r < n_gpus; r++) {
            const pid_t pid = fork();
            if (pid < 0) {
                fprintf(stderr, "error: fork failed\n");
                return 1;
            }
            if (pid == 0) {
                g_rank = r;
                kids.clear();
                verbose = false;
                dump_tokens = false;
                break;
            }
            kids.push_back(pid);
        }
    }
    auto finish = [&](int rc) -> int {  // children leave quietly; rank 0 reaps them and cleans up
        fflush(stdout);
        if (g_rank != 0) _exit(rc);
        for (pid_t k : kids) {
            int st = 0;
            waitpid(k, &st, 0);
            if (rc == 0 && !(WIFEXITED(st) && WEXITSTATUS(st) == 0)) rc = 1;
        }
        if (!xdir.empty()) {
            for (int r = 0; r < g_world; r++) {
                unlink((xdir + "/h" + std::to_string(r)).c_str());
                unlink((xdir + "/e" + std::to_string(r)).c_str());
            }
            rmdir(xdir.c_str());
        }
        return rc;
    };

    // ---- checkpoint: 28-byte header + f32 blob (:936-967); mmap instead of a heap copy:
    // l2z_weights_init streams it to the GPU once and the mapping is dropped
    const int fd = open(bin_path, O_RDONLY);
    if (fd < 0) {
        if (g_rank == 0) fprintf(stderr, "error: cannot open checkpoint '%s'\n", bin_path);
        return finish(1);
    }
    struct stat st;
    if (fstat(fd, &st) != 0 || (size_t)st.st_size < sizeof(l2z_config)) {
        if (g_rank == 0) fprintf(stderr, "error: checkpoint '%s' is too small\n", bin_path);
        return finish(1);
    }
    void *map = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (map == MAP_FAILED) {
        if (g_rank == 0) fprintf(stderr, "error: mmap of '%s' failed\n", bin_path);
        return finish(1);
    }
    l2z_config cfg;
    memcpy(&cfg, map, sizeof cfg);                       // :941
    const bool shared_weights = cfg.vocab_size > 0;      // :943
    cfg.vocab_size = abs(cfg.vocab_size);                // :944
    LOGV("config: dim %d hidden_dim %d n_layers %d n_heads %d n_kv_heads %d vocab_size %d seq_len %d\n",
         cfg.dim, cfg.hidden_dim, cfg.n_layers, cfg.n_heads, cfg.n_kv_heads, cfg.vocab_size, cfg.seq_len);
    LOGV("shared weights: %s\ntemperature: %g\ntop-p: %g\n", shared_weights ? "true" : "false",
         temperature, top_p);
    int device = 0;
    {
        char name[256] = "";
        int cus = 0, n_dev = 0;
        uint64_t hbm = 0;
        if (l2z_device_count(&n_dev) != L2Z_OK || n_dev < 1) return finish(die("no GPU"));
        device = g_rank % n_dev;  // fewer GPUs than ranks: ranks share (functional, not fast)
        if (g_world > n_dev) {
            // Ranks sharing a chip: a mat-vec that polls for a peer's words must leave the peer's producer
            // room to run, else the waits time out (at most 512 polling blocks on the chip in
            // total).  The library reads its knobs once, at first use -- nothing has used them yet.
            const int per_dev = (g_world + n_dev - 1) / n_dev;
            const int cap = 512 / per_dev < 32 ? 32 : 512 / per_dev;
            setenv("L2Z_GRID_CAP", std::to_string(cap).c_str(), 0);
        }
        if (l2z_device_info(device, name, sizeof name, &cus, &hbm) != L2Z_OK) return finish(die("no GPU"));
        LOGV("device: %s, %d CUs, %.0f GB HBM%s\n\n", name, cus, (double)hbm / 1e9,
             g_world > 1 ? " (rank 0 of the shard group)" : "");
    }
    l2z_comm *comm = nullptr;
    if (g_world > 1) {
        if (l2z_comm_init(g_rank, g_world, nullptr, device, &comm) != L2Z_OK) {
            mark_failed(xdir);
            return finish(die("comm_init"));
        }
        char handle[L2Z_COMM_IPC_BYTES];
        // (world * dim: under L2Z_SCHEME_B every rank's whole partial [dim] vector lands in every slot)
        const size_t longest = std::max((size_t)std::max(std::max(cfg.dim, cfg.hidden_dim), cfg.vocab_size),
                                        (size_t)g_world * (size_t)cfg.dim);
        const size_t widest = (size_t)std::max(cfg.dim, cfg.hidden_dim);  // bulk regions: [chunk, dim | hidden_dim]
        std::vector<char> all;
        if (l2z_comm_p2p_export_sized(comm, longest, widest, handle) != L2Z_OK) {
            mark_failed(xdir);
            return finish(die("p2p_export"));
        }
        if (!exchange_handles(xdir, kids, handle, &all)) {
            mark_failed(xdir);
            fprintf(stderr, "error: rank %d: hand-shake with the other ranks failed (a rank could not set up "
                            "its GPU, or died)\n", g_rank);
            return finish(1);
        }
        if (l2z_comm_p2p_connect(comm, all.data()) != L2Z_OK) return finish(die("p2p_connect"));
    }
    const float *data = reinterpret_cast<const float *>(static_cast<const char *>(map) + sizeof cfg);
    const size_t n_floats = ((size_t)st.st_size - sizeof cfg) / sizeof(float);
    l2z_weights *w = nullptr;
    const auto t_up = std::chrono::steady_clock::now();
    if (l2z_weights_init(&cfg, data, n_floats, shared_weights, comm, &w) != L2Z_OK)
        return finish(die("Weights.init"));
    {
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_up).count();
        LOGV("weights: %.2f GB of the file%s on the GPU in %.2f s\n", (double)st.st_size / 1e9 / g_world,
             g_world > 1 ? " (this rank's rows)" : "", el);
    }
    munmap(map, (size_t)st.st_size);
    close(fd);

    Tokenizer tok;  // :970
    std::string err;
    if (!tok.from_file(tokenizer_path, (size_t)cfg.vocab_size, &err)) {
        if (g_rank == 0) fprintf(stderr, "error: %s\n", err.c_str());
        return finish(1);
    }
    l2z_runstate *s = nullptr;  // :974
    if (l2z_runstate_init(&cfg, comm, &s) != L2Z_OK) return finish(die("RunState.init"));

    std::vector<int32_t> prompt;  // :978-985
    if (input && !tok.encode(input, &prompt, &err)) {
        if (g_rank == 0) fprintf(stderr, "error: cannot encode the prompt: %s\n", err.c_str());
        return finish(1);
    }
    const size_t prompt_len = prompt.size();

    seq_len = seq_len == 0 ? (size_t)cfg.seq_len : seq_len;                        // :992
    seq_len = seq_len < 1 ? 1 : (seq_len > (size_t)cfg.seq_len ? (size_t)cfg.seq_len : seq_len);  // :993

    std::vector<float> logits((size_t)cfg.vocab_size);
    std::vector<IndexedF32> logits_indexed;
    std::vector<int32_t> produced;
    bool timer_started = false;
    std::chrono::steady_clock::time_point t0;
    size_t token = 1, pos = 0;  // :988, :994

    // print one token exactly as :1022-1041 does; returns false if the sequence ended
    auto emit = [&](size_t next) -> bool {
        produced.push_back((int32_t)next);
        if (next == 1) return false;  // :1017
        std::string_view piece = tok.tokens[next];
        if (token == 1 && !piece.empty() && piece[0] == ' ') piece.remove_prefix(1);  // :1022-1025
        const int byte = is_raw_byte(piece);
        if (byte >= 0) {  // :1028-1031: printed, but the timer is not started on this path
            if (g_rank == 0) fputc(byte, stdout);
            token = next;
            return true;
        }
        if (g_rank == 0) fwrite(piece.data(), 1, piece.size(), stdout);
        token = next;
        if (!timer_started) {  // :1039-1041
            fflush(stdout);
            timer_started = true;
            t0 = std::chrono::steady_clock::now();
        }
        return true;
    };

    if (temperature == 0.0f) {
        // the loop of :995-1042 on the device; prompt override included
        if (l2z_greedy_begin(s, prompt.data(), (int)prompt_len) != L2Z_OK) return finish(die("greedy_begin"));
        std::vector<int32_t> chunk(64);
        bool alive = true;
        // tokens per call: the text appears in bursts of one call, so keep a call near 40 ms -- 64 tokens for
        // the small models, ~9 for the 7B shape (a call costs one stream synchronisation, ~20 us)
        size_t step = 8;
        if (g_world > 1) {
            // The ranks share no control plane: they stay in step only because every rank makes the SAME
            // sequence of calls.  A step sized from this rank's own clock can differ between ranks (9 vs 8
            // tokens), and when a BOS then ends the sequence inside a call the rank that asked for more has
            // queued passes its peers never run (a 20 s gather timeout, stores into a freed arena).  So the
            // step is a function of the model and the rank count only: this rank's weight bytes per token at
            // ~5 TB/s plus ~5 us per gather, sized for the same ~40 ms bursts.
            const double wbytes = 4.0 * ((double)cfg.n_layers * (2.0 * cfg.dim * cfg.dim +
                                                                2.0 * cfg.dim * (cfg.dim / cfg.n_heads) * cfg.n_kv_heads +
                                                                3.0 * cfg.dim * cfg.hidden_dim) +
                                         (double)cfg.vocab_size * cfg.dim) / g_world;
            const double per = wbytes / 5.0e12 + (4.0 * cfg.n_layers + 1.0) * 5.0e-6;
            const double fit = 0.040 / per;
            step = fit < 4.0 ? 4 : fit > 64.0 ? 64 : (size_t)fit;
        }
        while (alive && pos < seq_len) {
            // first token alone so the clock starts where the reference starts it; a prompt of
            // L2Z_PREFILL_MIN_PROMPT tokens or more is asked for in one call so that the library
            // runs its positions as one batched pass (they then print in a burst)
            int want = pos == 0 ? 1 : (int)std::min<size_t>(step, seq_len - pos);
            if (pos == 0 && prompt_len >= L2Z_PREFILL_MIN_PROMPT && prompt_len <= seq_len) {
                want = (int)prompt_len;
                chunk.resize(std::max(chunk.size(), prompt_len));
            }
            int got = 0;
            const auto tc = std::chrono::steady_clock::now();
            if (l2z_greedy_run(&cfg, s, w, want, chunk.data(), &got) != L2Z_OK) return finish(die("greedy_run"));
            if (got == 0) break;
            if (g_world == 1 && pos > 0 && got == want && want >= 4) {
                const double per = std::chrono::duration<double>(std::chrono::steady_clock::now() - tc).count() / got;
                const double fit = per > 0.0 ? 0.040 / per : 64.0;
                step = fit < 4.0 ? 4 : fit > 64.0 ? 64 : (size_t)fit;
            }
            for (int i = 0; i < got && alive; i++) {
                alive = emit((size_t)chunk[(size_t)i]);
                if (alive) pos++;
            }
        }
    } else {
        // The prompt positions (:999-1000: next = prompt[pos], their logits are never sampled) as
        // one batched pass when the prompt is long enough; the tokens print in a burst, then the
        // loop continues at pos = prompt_len.  Any refusal (odd dims, L2Z_PREFILL=0, a BOS inside
        // the prompt, which would end the loop at :1017) leaves the stepped loop below to do it.
        // L2Z_HOST_SOFTMAX=1: divide + softmax on the host in the reference's order (:1005-1008) instead of
        // l2z_probs_read -- the device reduces the denominator as a tree, so probabilities can differ in the
        // last ulp and a fixed seed can (rarely) land on the other side of a cdf boundary
        const char *hs_env = getenv("L2Z_HOST_SOFTMAX");
        const bool host_softmax = hs_env && atoi(hs_env) != 0;
        const char *pf_env = getenv("L2Z_PREFILL");
        bool has_bos = false;
        for (int32_t t : prompt) has_bos = has_bos || t == 1;
        if (prompt_len >= L2Z_PREFILL_MIN_PROMPT && prompt_len <= seq_len && !has_bos &&
            !(pf_env && atoi(pf_env) == 0)) {  // (a shard group without a bulk transport refuses: stepped loop)
            std::vector<int32_t> in(prompt_len);
            in[0] = 1;
            for (size_t i = 1; i < prompt_len; i++) in[i] = prompt[i - 1];
            if (l2z_prefill(in.data(), (int)prompt_len, 0, &cfg, s, w) == L2Z_OK) {
                for (size_t i = 0; i < prompt_len; i++) emit((size_t)prompt[i]);
                pos = prompt_len;
            }
        }
        for (; pos < seq_len; pos++) {
            if (l2z_transformer((int)token, (int)pos, &cfg, s, w) != L2Z_OK) return finish(die("transformer"));  // :996
            size_t next;
            if (pos < prompt_len) {
                next = (size_t)prompt[pos];  // :999-1000
            } else {
                // :1005-1008 (logits / temperature, softmax) on the device, then the copy the samplers need
                // anyway: 32000 exp() on one host core take as long as a small model's forward pass
                if (host_softmax) {
                    if (l2z_logits_read(s, logits.data()) != L2Z_OK) return finish(die("logits_read"));
                    for (float &v : logits) v /= temperature;  // :1006
                    softmax(logits.data(), logits.size());      // :1008
                } else if (l2z_probs_read(s, temperature, logits.data()) != L2Z_OK) {
                    return finish(die("probs_read"));
                }
                next = (top_p == 0.0f || top_p == 1.0f)         // :1009-1012
                           ? sample(logits.data(), logits.size(), prng)
                           : sample_top_p(logits.data(), logits.size(), top_p, logits_indexed, prng);
            }
            if (!emit(next)) break;
        }
    }
    fflush(stdout);
    if (timer_started) {  // :1043-1050 (the reference panics when no token was ever printed)
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const double tps = pos >= 1 ? (double)(pos - 1) / el : 0.0;
        LOGV("\n\n%u tokens per second\n", (unsigned)tps);
    }
    if (dump_tokens) {
        fprintf(stderr, "tokens:");
        for (int32_t t : produced) fprintf(stderr, " %d", t);
        fprintf(stderr, "\n");
    }
    l2z_runstate_free(s);
    l2z_weights_free(w);
    if (comm) l2z_comm_free(comm);
    return finish(0);
}
