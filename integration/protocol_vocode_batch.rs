// Source of the `hip-vocoder` feature's Rust side.  Written against the reference tree; NOT compiled
// here (this image has no cargo / rustc).
//
// goes into src/protocol.rs.  Both variants are APPENDED -- after `Ping` in `Request` (protocol.rs:14-68) and after `Init`
// in `ResponseData` (protocol.rs:72-107).  rmp-serde tags enum variants by NAME (externally tagged: {"Vocode": [...]}), so
// appending moves no existing encoding; an old worker answers a VocodeBatch with its "Deserialize request ... failed"
// error (mod.rs:673-674) and closes, which `Pipeline` below treats as "this worker cannot batch" and falls back.
// Known-answer bytes: tests/golden/wire_kat.json (request_vocode_batch_*, response_ok_vocode_batch_*), asserted against
// rmp-serde by integration/wire_kat_test.rs and against the C codec by tests/test_wire_cpu.py.

// enum Request { ...existing variants...,
    /// Convert several code sequences to audio in one batched decode (hip-vocoder workers).
    VocodeBatch {
        /// Per sequence: raw int64 little-endian bytes [N_i, 16]
        codes: Vec<Vec<u8>>,
        n_tokens: Vec<usize>,
    },
// }

// enum ResponseData { ...existing variants...,
    /// VocodeBatch response: per sequence what `Vocode` carries, in request order
    VocodeBatch {
        /// Per sequence: raw i16 little-endian PCM bytes
        audio: Vec<Vec<u8>>,
        n_samples: Vec<usize>,
    },
// }

// and in src/worker/mod.rs read_request's method_name match (mod.rs:676-685):
//     Request::VocodeBatch { .. } => "vocode_batch",
