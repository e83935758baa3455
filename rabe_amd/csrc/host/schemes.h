// rabe::schemes::* in C++: the reference's scheme API (same function names, argument meaning and error
// behaviour) with the group arithmetic executed by the HIP engine through the C ABI.
//
//   ac17  src/schemes/ac17/mod.rs:58-430   setup / cp_keygen / cp_encrypt / cp_decrypt   (+ *_batch)
//   bsw   src/schemes/bsw/mod.rs:39-318    setup / keygen / encrypt / decrypt
//   lsw   src/schemes/lsw/mod.rs:40-290    setup / keygen / encrypt / decrypt
//   aw11  src/schemes/aw11/mod.rs:46-390   setup / authgen / keygen / encrypt / decrypt
//   bdabe src/schemes/bdabe/mod.rs:49-475, mke08 src/schemes/mke08/mod.rs:20-470   the DNF-policy schemes
// Struct fields mirror the reference structs one for one.  Every function takes the Engine (GPU context) and,
// where the reference draws from thread_rng(), an Rng, as its first arguments; the rest is the reference signature.
#pragma once
#include "common.h"

namespace rabe { namespace schemes {

typedef std::pair<std::string, PolicyLanguage> PolicyRef;
// per item of a *_decrypt_batch: ok flag + plaintext or error text (a non-matching key fails its item, not the batch)
struct DecryptResult { bool ok; Bytes plaintext; std::string error; };

namespace ac17 {
struct Ac17PublicKey { G1 g; std::vector<G2> h_a; std::vector<Gt> e_gh_ka; };                  // :62-66
struct Ac17MasterKey { G1 g; G2 h; std::vector<G1> g_k; std::vector<Fr> a; std::vector<Fr> b; };   // :72-78
struct Ac17Ciphertext { std::vector<G2> c_0; std::vector<std::pair<std::string, std::vector<G1>>> c; Gt c_p; Bytes ct; };   // :84-89
struct Ac17CpCiphertext { PolicyRef policy; Ac17Ciphertext ct; };                               // :95-98
struct Ac17SecretKey { std::vector<G2> k_0; std::vector<std::pair<std::string, std::vector<G1>>> k; std::vector<G1> k_p; };   // :113-117
struct Ac17CpSecretKey { std::vector<std::string> attr; Ac17SecretKey sk; };                    // :132-135
struct Ac17KpCiphertext { std::vector<std::string> attr; Ac17Ciphertext ct; };                  // :104-107
struct Ac17KpSecretKey { PolicyRef policy; Ac17SecretKey sk; };                                 // :123-126

std::pair<Ac17PublicKey, Ac17MasterKey> setup(Engine& eng, Rng& rng);
Ac17CpSecretKey cp_keygen(Engine& eng, Rng& rng, const Ac17MasterKey& msk, const std::vector<std::string>& attributes);
Ac17CpCiphertext cp_encrypt(Engine& eng, Rng& rng, const Ac17PublicKey& pk, const std::string& policy, const Bytes& plaintext,
                            PolicyLanguage language);
Bytes cp_decrypt(Engine& eng, const Ac17CpSecretKey& sk, const Ac17CpCiphertext& ct);
// n independent calls in one engine launch set (the reason the engine exists); element i of the result
// equals cp_encrypt(pk, policies[i], plaintexts[i]) with the randomness drawn item after item.
std::vector<Ac17CpCiphertext> cp_encrypt_batch(Engine& eng, Rng& rng, const Ac17PublicKey& pk, const std::vector<std::string>& policies,
                                               const std::vector<Bytes>& plaintexts, PolicyLanguage language);
typedef schemes::DecryptResult DecryptResult;
void msk_tables(Engine& eng, const Ac17MasterKey& msk, rhip_g1_table** g, rhip_g2_table** h);     // packed.cpp: cached window tables of msk.g / msk.h
// n keys under one master key in one call (packed.cpp): item i gets the attribute list sets[item_set[i]]; records = Ac17CpSecretKey
bool cp_keygen_packed(Engine& eng, Rng& rng, const Ac17MasterKey& msk, const std::vector<std::vector<std::string>>& sets, size_t n,
                      const uint32_t* item_set, uint8_t* out_buf, size_t out_cap, uint64_t* out_off);
// packed forms: n ciphertexts = one blob of canonical records + offsets (schemes.cpp: "packed batches")
bool cp_encrypt_packed(Engine& eng, Rng& rng, const Ac17PublicKey& pk, const std::vector<std::string>& policies, PolicyLanguage language, size_t n,
                       const uint32_t* item_policy, const uint8_t* pt_blob, const uint64_t* pt_off, uint8_t* out_buf, size_t out_cap, uint64_t* out_off);
bool cp_decrypt_packed(Engine& eng, const Ac17CpSecretKey& sk, size_t n, const uint8_t* ct_blob, size_t ct_len, const uint64_t* ct_off, bool trusted,
                       int32_t* status, uint8_t* pt_buf, size_t pt_cap, uint64_t* pt_off, std::vector<std::string>* errors);
std::vector<DecryptResult> cp_decrypt_batch(Engine& eng, const std::vector<const Ac17CpSecretKey*>& sks,
                                            const std::vector<const Ac17CpCiphertext*>& cts);
// the Gt value handed to decrypt_symmetric (parity hook for tests; not part of the reference API)
Gt cp_decrypt_gt(Engine& eng, const Ac17CpSecretKey& sk, const Ac17CpCiphertext& ct);
// KP-ABE variant (:439-675), same kernels
Ac17KpSecretKey kp_keygen(Engine& eng, Rng& rng, const Ac17MasterKey& msk, const std::string& policy, PolicyLanguage lang);
// n calls of kp_keygen under one master key in one launch set (the triple loop of ac17/mod.rs:479-538 as Fr work on all host cores + one
// fixed-base launch); records = Ac17KpSecretKey, written on the device.  Buffers / return value as cp_keygen_packed.
bool kp_keygen_packed(Engine& eng, Rng& rng, const Ac17MasterKey& msk, const std::vector<std::string>& policies, PolicyLanguage lang, size_t n,
                      const uint32_t* item_policy, uint8_t* out_buf, size_t out_cap, uint64_t* out_off);
Ac17KpCiphertext kp_encrypt(Engine& eng, Rng& rng, const Ac17PublicKey& pk, const std::vector<std::string>& attributes, const Bytes& data);
Bytes kp_decrypt(Engine& eng, const Ac17KpSecretKey& sk, const Ac17KpCiphertext& ct);
Gt kp_decrypt_gt(Engine& eng, const Ac17KpSecretKey& sk, const Ac17KpCiphertext& ct);
std::vector<Ac17KpCiphertext> kp_encrypt_batch(Engine& eng, Rng& rng, const Ac17PublicKey& pk, const std::vector<std::vector<std::string>>& attribute_sets,
                                               const std::vector<Bytes>& datas);
// packed forms of the KP pair (schemes.cpp: the CP entry points with the record head and the roles of the names swapped)
bool kp_encrypt_packed(Engine& eng, Rng& rng, const Ac17PublicKey& pk, const std::vector<std::vector<std::string>>& sets, size_t n, const uint32_t* item_set,
                       const uint8_t* pt_blob, const uint64_t* pt_off, uint8_t* out_buf, size_t out_cap, uint64_t* out_off);
bool kp_decrypt_packed(Engine& eng, const Ac17KpSecretKey& sk, size_t n, const uint8_t* ct_blob, size_t ct_len, const uint64_t* ct_off, bool trusted,
                       int32_t* status, uint8_t* pt_buf, size_t pt_cap, uint64_t* pt_off, std::vector<std::string>* errors);
std::vector<DecryptResult> kp_decrypt_batch(Engine& eng, const std::vector<const Ac17KpSecretKey*>& sks, const std::vector<const Ac17KpCiphertext*>& cts);
}  // namespace ac17

namespace bsw {
struct CpAbePublicKey { G1 g1; G2 g2; G1 h; G2 f; Gt e_gg_alpha; };            // :43-49
struct CpAbeMasterKey { Fr beta; G2 g2_alpha; };                                // :55-58
struct CpAbeAttribute { std::string string; G1 g1; G2 g2; };                    // :85-89
struct CpAbeCiphertext { PolicyRef policy; G1 c; Gt c_p; std::vector<CpAbeAttribute> c_y; Bytes data; };   // :64-70
struct CpAbeSecretKey { G2 d; std::vector<CpAbeAttribute> d_j; };               // :76-79

std::pair<CpAbePublicKey, CpAbeMasterKey> setup(Engine& eng, Rng& rng);
bool keygen(Engine& eng, Rng& rng, const CpAbePublicKey& pk, const CpAbeMasterKey& msk, const std::vector<std::string>& attributes,
            CpAbeSecretKey* out);     // Option<..>: false = None
bool delegate(Engine& eng, Rng& rng, const CpAbePublicKey& pk, const CpAbeSecretKey& sk, const std::vector<std::string>& subset,
              CpAbeSecretKey* out);   // :162-206, Option<..>
CpAbeCiphertext encrypt(Engine& eng, Rng& rng, const CpAbePublicKey& pk, const std::string& policy, PolicyLanguage language,
                        const Bytes& plaintext);
Bytes decrypt(Engine& eng, const CpAbeSecretKey& sk, const CpAbeCiphertext& ct);
Gt decrypt_gt(Engine& eng, const CpAbeSecretKey& sk, const CpAbeCiphertext& ct);
// n independent calls with the group work of all items concatenated into one launch per operation type
// (BASELINE config 3: batch 4096); element i equals encrypt(pk, policies[i], plaintexts[i]) / decrypt(sks[i], cts[i])
// with the randomness drawn item after item.
std::vector<CpAbeCiphertext> encrypt_batch(Engine& eng, Rng& rng, const CpAbePublicKey& pk, const std::vector<std::string>& policies,
                                           PolicyLanguage language, const std::vector<Bytes>& plaintexts);
std::vector<DecryptResult> decrypt_batch(Engine& eng, const std::vector<const CpAbeSecretKey*>& sks, const std::vector<const CpAbeCiphertext*>& cts);
// the same two batches with packed input and output (packed.cpp): one blob of canonical records + offsets per side, caller-allocated
// buffers, the device-resident Level B path (rhip_bsw_{encrypt,decrypt}_batch).  Conventions as ac17::cp_{encrypt,decrypt}_packed.
// n keys under one master key in one call (packed.cpp): item i gets the attribute list sets[item_set[i]]; records = CpAbeSecretKey
// n calls of bsw::delegate on one key (packed.cpp): item i delegates `sk` to subsets[item_subset[i]]; records = CpAbeSecretKey
bool delegate_packed(Engine& eng, Rng& rng, const CpAbePublicKey& pk, const CpAbeSecretKey& sk, const std::vector<std::vector<std::string>>& subsets, size_t n,
                     const uint32_t* item_subset, uint8_t* out_buf, size_t out_cap, uint64_t* out_off);
bool keygen_packed(Engine& eng, Rng& rng, const CpAbePublicKey& pk, const CpAbeMasterKey& msk, const std::vector<std::vector<std::string>>& sets, size_t n,
                   const uint32_t* item_set, uint8_t* out_buf, size_t out_cap, uint64_t* out_off);
bool encrypt_packed(Engine& eng, Rng& rng, const CpAbePublicKey& pk, const std::vector<std::string>& policies, PolicyLanguage language, size_t n,
                    const uint32_t* item_policy, const uint8_t* pt_blob, const uint64_t* pt_off, uint8_t* out_buf, size_t out_cap, uint64_t* out_off);
bool decrypt_packed(Engine& eng, const CpAbeSecretKey& sk, size_t n, const uint8_t* ct_blob, size_t ct_len, const uint64_t* ct_off, bool trusted,
                    int32_t* status, uint8_t* pt_buf, size_t pt_cap, uint64_t* pt_off, std::vector<std::string>* errors);
}  // namespace bsw

namespace lsw {
struct KpAbePublicKey { G1 g1; G2 g2; G1 g1_b; G1 g1_b2; G1 h_b; Gt e_gg_alpha; };      // :44-51
struct KpAbeMasterKey { Fr alpha1; Fr alpha2; Fr b; G1 h_g1; G2 h_g2; };                  // :57-63
struct KpAbeKeyRow { std::string name; G1 d1; G2 d2; G1 d3; G1 d4; G1 d5; };              // (String, G1, G2, G1, G1, G1) :71
struct KpAbeSecretKey { PolicyRef policy; std::vector<KpAbeKeyRow> dj; };                 // :69-72
struct KpAbeCtRow { std::string name; G1 e1; G1 e2; G1 e3; };                             // (String, G1, G1, G1) :81
struct KpAbeCiphertext { Gt e1; G2 e2; std::vector<KpAbeCtRow> ej; Bytes ct; };           // :78-83

std::pair<KpAbePublicKey, KpAbeMasterKey> setup(Engine& eng, Rng& rng);
KpAbeSecretKey keygen(Engine& eng, Rng& rng, const KpAbePublicKey& pk, const KpAbeMasterKey& msk, const std::string& policy,
                      PolicyLanguage language);
KpAbeCiphertext encrypt(Engine& eng, Rng& rng, const KpAbePublicKey& pk, const std::vector<std::string>& attributes, const Bytes& plaintext);
Bytes decrypt(Engine& eng, const KpAbeSecretKey& sk, const KpAbeCiphertext& ct);
Gt decrypt_gt(Engine& eng, const KpAbeSecretKey& sk, const KpAbeCiphertext& ct);
// BASELINE config 4 (keygen + decrypt, batch 16384): n independent calls, one launch per operation type
std::vector<KpAbeSecretKey> keygen_batch(Engine& eng, Rng& rng, const KpAbePublicKey& pk, const KpAbeMasterKey& msk,
                                         const std::vector<std::string>& policies, PolicyLanguage language);
std::vector<DecryptResult> decrypt_batch(Engine& eng, const std::vector<const KpAbeSecretKey*>& sks, const std::vector<const KpAbeCiphertext*>& cts);
// packed forms (packed.cpp) over rhip_lsw_{keygen,decrypt}_batch: n keys as one blob of KpAbeSecretKey records; decrypt takes the n keys
// and ONE ciphertext (BASELINE config 4).  Positive attributes only -- policies / selections with "!x" take the object API.
// n ciphertexts in one call (packed.cpp): item i under the attribute list sets[item_set[i]]; records = KpAbeCiphertext
bool encrypt_packed(Engine& eng, Rng& rng, const KpAbePublicKey& pk, const std::vector<std::vector<std::string>>& sets, size_t n, const uint32_t* item_set,
                    const uint8_t* pt_blob, const uint64_t* pt_off, uint8_t* out_buf, size_t out_cap, uint64_t* out_off);
bool keygen_packed(Engine& eng, Rng& rng, const KpAbePublicKey& pk, const KpAbeMasterKey& msk, const std::vector<std::string>& policies,
                   PolicyLanguage language, size_t n, const uint32_t* item_policy, uint8_t* out_buf, size_t out_cap, uint64_t* out_off);
bool decrypt_packed(Engine& eng, const KpAbeCiphertext& ct, size_t n, const uint8_t* sk_blob, size_t sk_len, const uint64_t* sk_off, bool trusted,
                    int32_t* status, uint8_t* pt_buf, size_t pt_cap, uint64_t* pt_off, std::vector<std::string>* errors);
}  // namespace lsw

namespace aw11 {
struct Aw11GlobalKey { G1 g1; G2 g2; };                                                    // :50-53
struct Aw11PkAttr { std::string name; Gt egg_alpha; G2 g2_y; };
struct Aw11PublicKey { std::vector<Aw11PkAttr> attr; };                                    // :59-61
struct Aw11MkAttr { std::string name; Fr alpha; Fr y; };
struct Aw11MasterKey { std::vector<Aw11MkAttr> attr; };                                    // :67-69
struct Aw11CtRow { std::string name; Gt c1; G2 c2; G2 c3; };
struct Aw11Ciphertext { PolicyRef policy; Gt c_0; std::vector<Aw11CtRow> c; Bytes ct; };   // :75-80
struct Aw11SecretKey { std::string gid; std::vector<std::pair<std::string, G1>> attr; };   // :86-89

Aw11GlobalKey setup(Engine& eng, Rng& rng);
bool authgen(Engine& eng, Rng& rng, const Aw11GlobalKey& gk, const std::vector<std::string>& attributes, Aw11PublicKey* pk,
             Aw11MasterKey* msk);     // Option<..>
Aw11SecretKey keygen(Engine& eng, const Aw11GlobalKey& gk, const Aw11MasterKey& msk, const std::string& name,
                     const std::vector<std::string>& attributes);
void add_to_attribute(Engine& eng, const Aw11GlobalKey& gk, const Aw11MasterKey& msk, const std::string& attribute, Aw11SecretKey* sk);
Aw11Ciphertext encrypt(Engine& eng, Rng& rng, const Aw11GlobalKey& gk, const std::vector<const Aw11PublicKey*>& pks, const std::string& policy,
                       PolicyLanguage language, const Bytes& data);
Bytes decrypt(Engine& eng, const Aw11GlobalKey& gk, const Aw11SecretKey& sk, const Aw11Ciphertext& ct);
Gt decrypt_gt(Engine& eng, const Aw11GlobalKey& gk, const Aw11SecretKey& sk, const Aw11Ciphertext& ct);
// BASELINE config 5 (batch 8192): n independent calls, one launch per operation type
std::vector<Aw11Ciphertext> encrypt_batch(Engine& eng, Rng& rng, const Aw11GlobalKey& gk, const std::vector<const Aw11PublicKey*>& pks,
                                          const std::vector<std::string>& policies, PolicyLanguage language, const std::vector<Bytes>& datas);
std::vector<DecryptResult> decrypt_batch(Engine& eng, const Aw11GlobalKey& gk, const std::vector<const Aw11SecretKey*>& sks,
                                         const std::vector<const Aw11Ciphertext*>& cts);
// packed forms (packed.cpp) over rhip_aw11_{encrypt,decrypt}_batch; conventions as ac17::cp_{encrypt,decrypt}_packed
// n keys issued by one authority in one call (packed.cpp): user gids[i] gets the attribute list sets[item_set[i]]; records = Aw11SecretKey
bool keygen_packed(Engine& eng, const Aw11GlobalKey& gk, const Aw11MasterKey& msk, const std::vector<std::string>& gids,
                   const std::vector<std::vector<std::string>>& sets, size_t n, const uint32_t* item_set, uint8_t* out_buf, size_t out_cap, uint64_t* out_off);
bool encrypt_packed(Engine& eng, Rng& rng, const Aw11GlobalKey& gk, const std::vector<const Aw11PublicKey*>& pks, const std::vector<std::string>& policies,
                    PolicyLanguage language, size_t n, const uint32_t* item_policy, const uint8_t* pt_blob, const uint64_t* pt_off, uint8_t* out_buf,
                    size_t out_cap, uint64_t* out_off);
bool decrypt_packed(Engine& eng, const Aw11GlobalKey& gk, const Aw11SecretKey& sk, size_t n, const uint8_t* ct_blob, size_t ct_len, const uint64_t* ct_off,
                    bool trusted, int32_t* status, uint8_t* pt_buf, size_t pt_cap, uint64_t* pt_off, std::vector<std::string>* errors);
}  // namespace aw11

namespace ghw11 {
struct Ghw11PublicKey { G1 g1; G2 g2; G1 g1_a; G2 g2_a; Gt e_gg_alpha; };                  // :19-28
struct Ghw11MasterKey { G2 g2_alpha; Ghw11PublicKey pk; };                                 // :30-36
struct Ghw11Attribute { std::string string; G2 k_x; };                                     // :52-57
struct Ghw11SecretKey { G2 k; G2 l; std::vector<Ghw11Attribute> attr_key; };               // :38-49
struct Ghw11TransformKey { G2 k_z; G2 l_z; std::vector<Ghw11Attribute> attr_key_z; };     // :59-66
struct Ghw11RetrieveKey { Fr z; };                                                         // :68-73
struct Ghw11CtRow { std::string name; G1 c; G1 d; };                                       // (String, G1, G1) :82
struct Ghw11Ciphertext { PolicyRef policy; Gt c; G1 c1; std::vector<Ghw11CtRow> ci_di; Bytes data; };   // :75-84
struct Ghw11TransformCiphertext { Gt c; Gt t; };                                           // :86-91

std::pair<Ghw11PublicKey, Ghw11MasterKey> setup(Engine& eng, Rng& rng);
bool keygen(Engine& eng, Rng& rng, const Ghw11PublicKey& pk, const Ghw11MasterKey& msk, const std::vector<std::string>& attributes,
            Ghw11SecretKey* out);       // Option<..>: false = None
std::pair<Ghw11TransformKey, Ghw11RetrieveKey> tkgen(Engine& eng, Rng& rng, const Ghw11SecretKey& sk);
Ghw11Ciphertext encrypt(Engine& eng, Rng& rng, const Ghw11PublicKey& pk, const std::string& policy, PolicyLanguage language, const Bytes& plaintext);
// the outsourced part: m + 2 pairings (one final exponentiation) and 2m G1 multiplications per ciphertext
Ghw11TransformCiphertext transform(Engine& eng, const Ghw11Ciphertext& ct, const Ghw11TransformKey& tk);
// "decrypt-as-a-service": n independent transforms in one launch set; ok[i] false (with errors[i]) where tk i does not satisfy ct i
std::vector<Ghw11TransformCiphertext> transform_batch(Engine& eng, const std::vector<const Ghw11Ciphertext*>& cts,
                                                      const std::vector<const Ghw11TransformKey*>& tks, std::vector<std::string>* errors);
// packed form (packed.cpp): n ciphertext records in one blob + offsets, ONE transform key; out_buf + 768 i = Ghw11TransformCiphertext
// record (c | t) of item i, zeros where status[i] = -1.  The device-resident path: every Miller loop replays the key's prepared lines.
bool transform_packed(Engine& eng, const Ghw11TransformKey& tk, size_t n, const uint8_t* ct_blob, size_t ct_len, const uint64_t* ct_off, bool trusted,
                      int32_t* status, uint8_t* out_buf, size_t out_cap, std::vector<std::string>* errors);
Gt decrypt_out_gt(Engine& eng, const Ghw11TransformCiphertext& pct, const Ghw11RetrieveKey& rk);
Bytes decrypt_out(Engine& eng, const Ghw11TransformCiphertext& pct, const Ghw11RetrieveKey& rk, const Bytes& data);
}  // namespace ghw11

namespace bdabe {       // src/schemes/bdabe/mod.rs (DNF policies, multi-authority)
struct BdabePublicKey { G1 g1; G2 g2; G1 p1; G2 p2; Gt e_gg_y; };                          // :49-55
struct BdabeMasterKey { Fr y; };                                                            // :61-63
struct BdabeSecretUserKey { G1 u1; G2 u2; };                                                // :89-92
struct BdabePublicUserKey { std::string u; G1 u1; G2 u2; };                                 // :79-83
struct BdabeSecretAttributeKey { std::string attr; G1 au1; G2 au2; };                       // :98-102
struct BdabeUserKey { BdabeSecretUserKey sk; BdabePublicUserKey pk; std::vector<BdabeSecretAttributeKey> sk_a; };   // :69-73
struct BdabePublicAttributeKey { std::string attr; G1 a1; G2 a2; Gt a3; };                  // :108-113
struct BdabeSecretAuthorityKey { std::string name; G1 a1; G2 a2; Fr a3; };                  // :119-124
struct BdabeCiphertextTuple { std::vector<std::string> attr; Gt e1; G1 e2; G2 e3; G1 e4; G2 e5; };   // :130-137
struct BdabeCiphertext { PolicyRef policy; std::vector<BdabeCiphertextTuple> j; Bytes ct; };          // :143-147

std::pair<BdabePublicKey, BdabeMasterKey> setup(Engine& eng, Rng& rng);
BdabeSecretAuthorityKey authgen(Engine& eng, Rng& rng, const BdabePublicKey& pk, const BdabeMasterKey& msk, const std::string& name);
BdabeUserKey keygen(Engine& eng, Rng& rng, const BdabePublicKey& pk, const BdabeSecretAuthorityKey& ska, const std::string& name);
BdabePublicAttributeKey request_attribute_pk(Engine& eng, const BdabePublicKey& pk, const BdabeSecretAuthorityKey& ska, const std::string& attribute);
BdabeSecretAttributeKey request_attribute_sk(Engine& eng, const BdabePublicUserKey& pk_u, const BdabeSecretAuthorityKey& ska, const std::string& attribute);
BdabeCiphertext encrypt(Engine& eng, Rng& rng, const BdabePublicKey& pk, const std::vector<const BdabePublicAttributeKey*>& attr_pks,
                        const std::string& policy, PolicyLanguage language, const Bytes& plaintext);
Bytes decrypt(Engine& eng, const BdabeUserKey& sk, const BdabeCiphertext& ct);
Gt decrypt_gt(Engine& eng, const BdabeUserKey& sk, const BdabeCiphertext& ct);
// n independent decrypts: every item's four pairing factors on one accumulator, one launch set for the batch
std::vector<DecryptResult> decrypt_batch(Engine& eng, const std::vector<const BdabeUserKey*>& sks, const std::vector<const BdabeCiphertext*>& cts);
}  // namespace bdabe

namespace mke08 {       // src/schemes/mke08/mod.rs (DNF policies, multi-authority)
struct Mke08PublicKey { G1 g1; G2 g2; G1 p1; G2 p2; Gt e_gg_y1; Gt e_gg_y2; };              // :20-27
struct Mke08MasterKey { G1 g1; G2 g2; };                                                    // :32-35
struct Mke08SecretUserKey { G1 g1; G2 g2; };                                                // :58-61
struct Mke08PublicUserKey { std::string name; G1 g1; G2 g2; };                              // :49-53
struct Mke08SecretAttributeKey { std::string attr; G1 g1; G2 g2; };                         // :85-89
struct Mke08UserKey { Mke08SecretUserKey sk; Mke08PublicUserKey pk; std::vector<Mke08SecretAttributeKey> sk_a; };   // :40-44
struct Mke08SecretAuthorityKey { std::string name; Fr r; };                                 // :66-69
struct Mke08PublicAttributeKey { std::string attr; G1 g1; G2 g2; Gt gt1; Gt gt2; };         // :74-80
struct Mke08CTConjunction { std::vector<std::string> str; Gt j1; Gt j2; G1 j3; G2 j4; G1 j5; G2 j6; };   // :103-111
struct Mke08Ciphertext { PolicyRef policy; std::vector<Mke08CTConjunction> e; Bytes ct; };               // :94-98

std::pair<Mke08PublicKey, Mke08MasterKey> setup(Engine& eng, Rng& rng);
Mke08UserKey keygen(Engine& eng, Rng& rng, const Mke08PublicKey& pk, const Mke08MasterKey& msk, const std::string& name);
Mke08SecretAuthorityKey authgen(Rng& rng, const std::string& name);
Mke08PublicAttributeKey request_authority_pk(Engine& eng, const Mke08PublicKey& pk, const std::string& attribute, const Mke08SecretAuthorityKey& ska);
Mke08SecretAttributeKey request_authority_sk(Engine& eng, const Mke08PublicUserKey& pk_u, const std::string& attr, const Mke08SecretAuthorityKey& ska);
Mke08Ciphertext encrypt(Engine& eng, Rng& rng, const Mke08PublicKey& pk, const std::vector<const Mke08PublicAttributeKey*>& attr_pks,
                        const std::string& policy, PolicyLanguage language, const Bytes& plaintext);
Bytes decrypt(Engine& eng, const Mke08UserKey& sk, const Mke08Ciphertext& ct);
Gt decrypt_gt(Engine& eng, const Mke08UserKey& sk, const Mke08Ciphertext& ct);
std::vector<DecryptResult> decrypt_batch(Engine& eng, const std::vector<const Mke08UserKey*>& sks, const std::vector<const Mke08Ciphertext*>& cts);
}  // namespace mke08

}}  // namespace rabe::schemes

// pipelined packed batches (pipeline.cpp): the packed entry points above run on chunks of the items, a few chunks at a time, each on its
// own engine lane; results are those of the unchunked call.  `call(lo, hi, ...)` is the entry point bound to items [lo, hi).
#include <functional>
namespace rabe { namespace pipeline {
// `eng`: the engine the chunk runs on -- the host's own, or the member of a device GROUP (rabe_host_open_group) that owns the block
typedef std::function<bool(Engine& eng, size_t lo, size_t hi, Rng& rng, uint8_t* out, size_t cap, uint64_t* off)> ProduceFn;
typedef std::function<bool(Engine& eng, size_t lo, size_t hi, int32_t* status, uint8_t* pt, size_t cap, uint64_t* pt_off, std::vector<std::string>* errors)> ConsumeFn;
// engines: one entry = a plain host (min_chunk / RABE_PACKED_CHUNK / RABE_PACKED_LANES decide about chunks on its lanes, default: none);
// several = a device group: the items are cut into contiguous blocks, one per engine (sizes differ by at most one, in engine order),
// every block runs on its own thread with its engine's device current.  Outputs land where the unsplit call puts them and an ordered
// randomness source is drawn block after block -- the bytes do not depend on the split.
// min_chunk: fewest items a chunk may hold (RABE_PACKED_CHUNK overrides); RABE_PACKED_LANES: chunks in flight (default 2, 1 = unchunked)
bool produce(const std::vector<Engine*>& engines, Rng& rng, size_t n, size_t min_chunk, const ProduceFn& call, uint8_t* out_buf, size_t out_cap, uint64_t* out_off);
bool consume(const std::vector<Engine*>& engines, size_t n, size_t min_chunk, const uint64_t* in_off, size_t in_len, const ConsumeFn& call, int32_t* status, uint8_t* pt_buf,
             size_t pt_cap, uint64_t* pt_off, std::vector<std::string>* errors);
}}  // namespace rabe::pipeline
