/* hipvoc_wire.h -- the reference's worker wire format for the Vocoder role, as C entry points.
 *
 * Frame (src/worker_client.rs:31-60, src/worker/mod.rs:656-698): u32 big-endian payload length, then the
 * payload = rmp-serde 1.3 `to_vec` (compact) encoding of `Request` / `Response` (src/protocol.rs:14-116):
 *   - a struct is a msgpack ARRAY of its fields in declaration order (Response = [status, data, error]);
 *   - an enum is externally tagged: a unit variant is the variant name as a str ("Ping"); a struct variant is a
 *     one-entry MAP { "Vocode": [codes, n_tokens] };
 *   - Vec<u8> is a msgpack ARRAY of integers (serde's default for Vec<u8>; not bin), Option::None is nil.
 * The decoder here is lenient where serde is (struct payloads as arrays or maps keyed by field name, byte
 * fields as array-of-int or bin); the encoder emits exactly the compact form above.
 * No Rust toolchain exists in the build image, so this is pinned to the msgpack spec and rmp-serde's documented
 * behaviour, cross-checked against an independent msgpack implementation in tests/ -- not against rmp-serde
 * itself ("wire parity unpinned").
 *
 * VocodeBatch is this build's extension, appended after the reference's variants so existing
 * encodings are unchanged:  Request::VocodeBatch { codes: Vec<Vec<u8>>, n_tokens: Vec<usize> }
 *                           ResponseData::VocodeBatch { audio: Vec<Vec<u8>>, n_samples: Vec<usize> }
 */
#ifndef HIPVOC_WIRE_H
#define HIPVOC_WIRE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define HIPVOC_WIRE_VOCODE 0        /* Request::Vocode            src/protocol.rs:51-56 */
#define HIPVOC_WIRE_PING 1          /* Request::Ping              src/protocol.rs:66-67 */
#define HIPVOC_WIRE_INIT 2          /* Request::Init              src/protocol.rs:15-16 */
#define HIPVOC_WIRE_OTHER 3         /* a well-formed request of another role (talker / predictor) */
#define HIPVOC_WIRE_VOCODE_BATCH 4  /* extension */
#define HIPVOC_WIRE_E_MALFORMED (-1)
#define HIPVOC_WIRE_E_CAP (-2)

/* Classifies one request payload.  For Vocode, copies the code bytes (i64 LE, [N,16]) to codes_out and sets
 * *codes_len / *n_tokens; codes_out may be NULL to query *codes_len.  Replaces the rmp_serde::from_slice +
 * match of read_request / dispatch (src/worker/mod.rs:260-300, 673-685) for this role. */
int hipvoc_wire_parse_request(const uint8_t *msg, size_t len, uint8_t *codes_out, size_t codes_cap, size_t *codes_len,
                              uint64_t *n_tokens);

/* VocodeBatch: concatenated code bytes to codes_out, offsets[B+1] (byte offsets), n_tokens[B]. */
int hipvoc_wire_parse_vocode_batch(const uint8_t *msg, size_t len, uint8_t *codes_out, size_t codes_cap,
                                   size_t *offsets, uint64_t *n_tokens, size_t max_items, size_t *n_items);

/* Encoders return the payload size; nothing is written beyond cap (call with cap 0 to size the buffer).
 * Response{status:"ok", data:Vocode{audio: i16 LE bytes, n_samples}, error:None}   src/worker/mod.rs:525-530 */
size_t hipvoc_wire_encode_vocode(const int16_t *pcm, size_t n_samples, uint8_t *out, size_t cap);
size_t hipvoc_wire_encode_vocode_batch(const int16_t *pcm, const size_t *sample_offsets /*[B+1]*/, size_t n_items,
                                       uint8_t *out, size_t cap);
/* Response with a unit data variant: ok -> {"ok", variant, nil}; error -> {"error", "Init", message}
 * (src/worker/mod.rs:246-258, 702-708) */
size_t hipvoc_wire_encode_simple(int ok, const char *variant, const char *error, uint8_t *out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
