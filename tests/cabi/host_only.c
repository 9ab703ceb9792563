/* A plain-C caller of include/nar_fs2.h (compiled with gcc, -std=c99 -pedantic): the header must be usable from C, every
 * host-side entry point must be callable through dlopen/dlsym without a GPU, and the config struct's layout must be what
 * the Python binding (smart_nar_fast_tts_amd/_lib.py NsConfig) assumes.  Run by tests/test_cabi_and_host.py. */
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include "nar_fs2.h"

typedef const char* (*last_error_fn)(void);
typedef int (*create_fn)(const ns_config*, ns_model**);
typedef void (*destroy_fn)(ns_model*);
typedef size_t (*arena_fn)(const ns_model*);
typedef size_t (*ws_fn)(const ns_model*, int, int);
typedef int (*plan_fn)(int, int, int, int, int32_t*);
typedef int (*split_fn)(int, int, int, int);
typedef int (*version_fn)(void);

int main(int argc, char** argv) {
  void* so;
  ns_config c;
  ns_model* m = 0;
  int32_t plan[8];
  if (argc < 2) return 2;
  so = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!so) { printf("dlopen: %s\n", dlerror()); return 3; }
  {
    /* (POSIX idiom: ISO C has no conversion from void* to a function pointer) */
    last_error_fn last_error; create_fn create; destroy_fn destroy; arena_fn arena; ws_fn enc_ws; plan_fn plan_gemm; split_fn att_split; version_fn abi_version;
    *(void**)(&last_error) = dlsym(so, "ns_last_error");
    *(void**)(&create) = dlsym(so, "ns_create");
    *(void**)(&destroy) = dlsym(so, "ns_destroy");
    *(void**)(&arena) = dlsym(so, "ns_arena_bytes");
    *(void**)(&enc_ws) = dlsym(so, "ns_encoder_ws_bytes");
    *(void**)(&plan_gemm) = dlsym(so, "ns_plan_gemm");
    *(void**)(&att_split) = dlsym(so, "ns_plan_attention_split");
    *(void**)(&abi_version) = dlsym(so, "ns_abi_version");
    /* a caller built against this header refuses a library with another contract (round 5 grew ns_plan_gemm's out[] unversioned) */
    if (!abi_version || abi_version() != NS_ABI_VERSION) { printf("ABI version mismatch\n"); return 10; }
    if (!last_error || !create || !destroy || !arena || !enc_ws || !plan_gemm || !att_split) { printf("missing symbol\n"); return 4; }
    memset(&c, 0, sizeof(c));
    c.n_vocab = 361; c.max_seq_len = 1000;
    c.d_enc = 256; c.n_enc_layer = 4; c.n_enc_head = 2;
    c.d_dec = 256; c.n_dec_layer = 4; c.n_dec_head = 2;
    c.d_inner = 1024; c.ffn_k1 = 9; c.ffn_k2 = 1;
    c.vp_filter = 256; c.vp_kernel = 3; c.n_bins = 256; c.n_mel = 80;
    c.postnet_dim = 512; c.postnet_k = 5; c.postnet_n = 5;
    c.pitch_frame_level = 1; c.energy_frame_level = 1;
    if (create(&c, &m) != 0 || !m) { printf("ns_create: %s\n", last_error()); return 5; }
    printf("sizeof(ns_config)=%u arena_bytes=%lu enc_ws(16,128)=%lu\n", (unsigned)sizeof(ns_config), (unsigned long)arena(m), (unsigned long)enc_ws(m, 16, 128));
    if (arena(m) < 100u * 1000u * 1000u) return 6;
    c.d_dec = 512; /* encoder_hidden != decoder_hidden: refused with a message */
    { ns_model* bad = 0; if (create(&c, &bad) == 0 || strlen(last_error()) == 0) return 7; }
    if (plan_gemm(16160, 1024, 256, 9, plan) != 1 || plan[0] != 128 || plan[1] != 256 || plan[2] != 16160 || plan[5] != 0 || plan[6] != 32 || plan[7] <= 0) return 8;
    if (att_split(16, 1010, 2, 128) != 1) return 9;
    destroy(m);
  }
  printf("C caller ok\n");
  return 0;
}
