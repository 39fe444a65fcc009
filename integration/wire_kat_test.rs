//! Pins the wire codec of libhipvoc (include/hipvoc_wire.h, qwen3-tts_amd/csrc/wire.cpp) against rmp-serde ITSELF.
//!
//! This image has no cargo, so the C codec is tested against the independent Python `msgpack` package and against the
//! hand-assembled bytes of tests/golden/wire_kat.json.  A maintainer with a Rust toolchain drops this file into the
//! reference tree as `tests/wire_kat_test.rs` (it only needs the crate's own `protocol` module and `rmp_serde`,
//! both already dependencies: Cargo.toml) and runs `cargo test --test wire_kat_test`: if every assertion holds, the
//! bytes the C codec parses / emits are the bytes `worker_client.rs:31-60` sends / expects.
use qwen3_tts_rs::protocol::{Request, Response, ResponseData};

fn hex(b: &[u8]) -> String {
    b.iter().map(|x| format!("{:02x}", x)).collect()
}

#[test]
fn requests_serialize_to_the_known_answers() {
    // tests/golden/wire_kat.json
    assert_eq!(hex(&rmp_serde::to_vec(&Request::Ping).unwrap()), "a450696e67");
    assert_eq!(
        hex(&rmp_serde::to_vec(&Request::Init { role: "vocoder".into(), models_dir: "/models".into() }).unwrap()),
        "81a4496e697492a7766f636f646572a72f6d6f64656c73"
    );
    assert_eq!(
        hex(&rmp_serde::to_vec(&Request::Vocode { codes: vec![0x01, 0x80, 0xff], n_tokens: 0 }).unwrap()),
        "81a6566f636f6465929301cc80ccff00"
    );
    // the appended variant (integration/protocol_vocode_batch.rs)
    assert_eq!(
        hex(&rmp_serde::to_vec(&Request::VocodeBatch { codes: vec![vec![0x01, 0x80], vec![0xff]], n_tokens: vec![0, 0] }).unwrap()),
        "81ab566f636f6465426174636892929201cc8091ccff920000"
    );
}

#[test]
fn responses_deserialize_from_the_known_answers() {
    let unhex = |s: &str| -> Vec<u8> { (0..s.len()).step_by(2).map(|i| u8::from_str_radix(&s[i..i + 2], 16).unwrap()).collect() };
    let r: Response = rmp_serde::from_slice(&unhex("93a26f6ba450696e67c0")).unwrap();
    assert_eq!(r.status, "ok");
    assert!(matches!(r.data, ResponseData::Ping));
    assert!(r.error.is_none());
    let r: Response = rmp_serde::from_slice(&unhex("93a56572726f72a4496e6974a178")).unwrap();
    assert_eq!(r.status, "error");
    assert_eq!(r.error.as_deref(), Some("x"));
    let r: Response = rmp_serde::from_slice(&unhex("93a26f6b81a6566f636f6465929200cc8001c0")).unwrap();
    match r.data {
        ResponseData::Vocode { audio, n_samples } => {
            assert_eq!(audio, vec![0x00, 0x80]); // one sample, -32768, little-endian
            assert_eq!(n_samples, 1);
        }
        _ => panic!("expected ResponseData::Vocode"),
    }
    let r: Response = rmp_serde::from_slice(&unhex("93a26f6b81ab566f636f6465426174636892929200cc8090920100c0")).unwrap();
    match r.data {
        ResponseData::VocodeBatch { audio, n_samples } => {
            assert_eq!(audio, vec![vec![0x00, 0x80], vec![]]);
            assert_eq!(n_samples, vec![1, 0]);
        }
        _ => panic!("expected ResponseData::VocodeBatch"),
    }
    // and the other direction: what the Rust worker would have sent is what the C encoder emits
    let back = rmp_serde::to_vec(&Response { status: "ok".into(), data: ResponseData::Vocode { audio: vec![0x00, 0x80], n_samples: 1 }, error: None }).unwrap();
    assert_eq!(hex(&back), "93a26f6b81a6566f636f6465929200cc8001c0");
}
