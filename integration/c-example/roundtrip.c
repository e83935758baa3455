/* A C caller of the drop-in boundary (include/rabe_host.h + include/rabe_hip.h, plain C99): the reference's own test case
 * `and` (src/schemes/ac17/mod.rs:688-705) -- setup, cp_keygen, cp_encrypt, cp_decrypt -- then the same through the packed
 * entry points and one element-level call of the device ABI.  Build: see the Makefile next to this file.  Exit code 0 = every step gave
 * the expected result; without a HIP device rabe_host_create fails and the program says so (exit code 2). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rabe_hip.h"
#include "rabe_host.h"

#define CHECK(call)                                                                              \
  do {                                                                                           \
    int32_t rc_ = (call);                                                                        \
    if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #call, (int)rc_, rabe_host_last_error(h)); return 1; } \
  } while (0)

int main(void) {
  rabe_host* h = NULL;
  if (rabe_host_create(0, &h) != 0) {
    fprintf(stderr, "no usable HIP device: %s\n", rabe_host_last_error(NULL));
    return 2;
  }
  const char* plaintext = "dance like no one's watching, encrypt like everyone is!";
  const size_t len = strlen(plaintext);
  void *pk = NULL, *msk = NULL, *sk = NULL, *ct = NULL;
  CHECK(rabe_ac17_setup(h, &pk, &msk));
  const char* attrs[2] = {"A", "B"};
  CHECK(rabe_ac17_cp_keygen(h, msk, attrs, 2, &sk));
  CHECK(rabe_ac17_cp_encrypt(h, pk, "\"A\" and \"B\"", RABE_HUMAN_POLICY, (const uint8_t*)plaintext, len, &ct));
  uint8_t* out = NULL;
  size_t out_len = 0;
  CHECK(rabe_ac17_cp_decrypt(h, sk, ct, &out, &out_len));
  if (out_len != len || memcmp(out, plaintext, len) != 0) { fprintf(stderr, "object API: plaintext differs\n"); return 1; }
  rabe_bytes_free(out);

  /* the same two calls packed: 3 items, one blob of records each way, caller-allocated buffers */
  const char* policies[1] = {"\"A\" and \"B\""};
  const uint32_t item_policy[3] = {0, 0, 0};
  uint8_t pt_blob[3 * 64];
  uint64_t pt_off[4] = {0, 0, 0, 0};
  for (int i = 0; i < 3; i++) { memcpy(pt_blob + pt_off[i], plaintext, len); pt_off[i + 1] = pt_off[i] + len; }
  uint64_t ct_off[4];
  int32_t rc = rabe_ac17_cp_encrypt_packed(h, pk, policies, 1, RABE_HUMAN_POLICY, 3, item_policy, pt_blob, pt_off, NULL, 0, ct_off);
  if (rc != 1) { fprintf(stderr, "sizing call: expected 1, got %d\n", (int)rc); return 1; }        /* too small: offsets filled, nothing drawn */
  uint8_t* ct_buf = (uint8_t*)malloc((size_t)ct_off[3]);
  CHECK(rabe_ac17_cp_encrypt_packed(h, pk, policies, 1, RABE_HUMAN_POLICY, 3, item_policy, pt_blob, pt_off, ct_buf, (size_t)ct_off[3], ct_off));
  uint8_t* pt_buf = (uint8_t*)malloc((size_t)ct_off[3]);
  uint64_t out_off[4];
  int32_t status[3];
  CHECK(rabe_ac17_cp_decrypt_packed(h, sk, 3, ct_buf, (size_t)ct_off[3], ct_off, 0, status, pt_buf, (size_t)ct_off[3], out_off));
  for (int i = 0; i < 3; i++)
    if (status[i] != 0 || out_off[i + 1] - out_off[i] != len || memcmp(pt_buf + out_off[i], plaintext, len) != 0) {
      fprintf(stderr, "packed API: item %d differs\n", i);
      return 1;
    }
  free(ct_buf);
  free(pt_buf);

  /* one call of the device-level ABI (host-value form): 2 * (1, 2) on BN254's G1, the EIP-196 vector */
  rhip_ctx* ctx = NULL;
  if (rhip_ctx_create(0, &ctx) != RHIP_OK) { fprintf(stderr, "rhip_ctx_create: %s\n", rhip_last_error(NULL)); return 1; }
  rhip_g1 g, twice;
  rhip_fr two;
  memset(&g, 0, sizeof g);
  memset(&two, 0, sizeof two);
  g.l[0] = 1; g.l[8] = 2; two.l[0] = 2;
  if (rhip_host_g1_mul(ctx, &g, &two, &twice) != RHIP_OK) { fprintf(stderr, "rhip_host_g1_mul: %s\n", rhip_last_error(ctx)); return 1; }
  if (twice.l[7] != 0x030644e7u || twice.l[15] != 0x15ed738cu) { fprintf(stderr, "2 * (1, 2): unexpected point\n"); return 1; }
  rhip_ctx_destroy(ctx);

  rabe_obj_free(RABE_AC17_CP_CT, ct);
  rabe_obj_free(RABE_AC17_CP_SK, sk);
  rabe_obj_free(RABE_AC17_MSK, msk);
  rabe_obj_free(RABE_AC17_PK, pk);
  rabe_host_destroy(h);
  printf("ok\n");
  return 0;
}
