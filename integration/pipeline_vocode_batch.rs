// Source of the `hip-vocoder` feature's Rust side.  Written against the reference tree; NOT compiled
// here (this image has no cargo / rustc).
//
// Batched vocoder requests in the orchestrator.  Two things change in src/pipeline.rs and one in
// src/api/openai.rs / src/mcp/server.rs:
//
//  1. `VocoderQueue`: every piece of code that used to do `self.vocoder.send_request(&Request::Vocode{..})` ...
//     `self.vocoder.read_response()` (pipeline.rs:346-377, 586-607, 636-673) now calls `queue.submit(codes, n)` and later
//     awaits the returned receiver.  One task owns the vocoder connection; whenever the connection is free it takes
//     EVERYTHING that is waiting -- the 64-token chunks of one utterance, the text chunks of a pipelined synthesis, and
//     the chunks of OTHER concurrent utterances -- and sends it as ONE `Request::VocodeBatch`, so a single GPU is fed many
//     sequences at once.  With nothing else waiting a submission goes out alone at once: no latency is added.
//  2. The order of audio within one utterance is kept by the caller, which awaits its receivers in submission order
//     (exactly the "collect responses in submission order" loops it replaces).
//  3. `synthesis_lock` (api/openai.rs:86, mcp/server.rs:243) serialised whole syntheses because talker and predictor are
//     single-client state machines.  It now guards only phases 1-3 (tokenize, prefill, the talker/predictor loop); the
//     guard is dropped before the vocoder phase, whose requests are independent units (vocoder.rs:65-82) and are
//     multiplexed by the queue.  Utterance B's generation therefore overlaps utterance A's vocoding.

use tokio::sync::{mpsc, oneshot};

use crate::protocol::{Request, ResponseData};
use crate::worker_client::{decode_i16, WorkerClient};

struct Pending {
    codes: Vec<u8>,            // encode_i64 of [n, 16] codes (worker_client.rs:80-83)
    n_tokens: usize,
    reply: oneshot::Sender<anyhow::Result<Vec<i16>>>,
}

/// Handle to the task that owns the vocoder connection.  Cloneable: one per `Pipeline`, shared by concurrent requests.
#[derive(Clone)]
pub struct VocoderQueue {
    tx: mpsc::UnboundedSender<Pending>,
}

impl VocoderQueue {
    /// `max_batch`: sequences per VocodeBatch (the worker's own --max-items bounds it too).
    pub fn spawn(mut client: WorkerClient, max_batch: usize) -> Self {
        let (tx, mut rx) = mpsc::unbounded_channel::<Pending>();
        tokio::spawn(async move {
            let mut batching = true;                       // cleared when the worker turns out not to know VocodeBatch
            while let Some(first) = rx.recv().await {
                // everything that queued up while the previous batch was on the GPU
                let mut batch = vec![first];
                while batch.len() < max_batch {
                    match rx.try_recv() {
                        Ok(p) => batch.push(p),
                        Err(_) => break,
                    }
                }
                if batch.len() == 1 || !batching {
                    // a lone request keeps the reference's message (any worker understands it)
                    for p in batch {
                        let r = Self::one(&mut client, &p).await;
                        let _ = p.reply.send(r);
                    }
                    continue;
                }
                let req = Request::VocodeBatch {
                    codes: batch.iter().map(|p| p.codes.clone()).collect(),
                    n_tokens: batch.iter().map(|p| p.n_tokens).collect(),
                };
                match client.call(&req).await {
                    Ok(resp) => match resp.data {
                        ResponseData::VocodeBatch { audio, .. } if audio.len() == batch.len() => {
                            for (p, a) in batch.into_iter().zip(audio) {
                                let _ = p.reply.send(decode_i16(&a));
                            }
                        }
                        _ => {
                            for p in batch {
                                let _ = p.reply.send(Err(anyhow::anyhow!("Unexpected vocoder response")));
                            }
                        }
                    },
                    Err(e) => {
                        // An ONNX/RKNN worker of the unmodified reference cannot deserialize the new variant, answers
                        // with an error and closes (mod.rs:642-651): reconnect once, stop batching, replay one by one.
                        tracing::warn!("VocodeBatch failed ({}); falling back to single Vocode requests", e);
                        batching = false;
                        let addr = client.addr().to_string();
                        let reconnected = match addr.rsplit_once(':') {
                            Some((h, p)) => WorkerClient::connect(h, p.parse().unwrap_or(9092)).await,
                            None => Err(anyhow::anyhow!("bad vocoder address {}", addr)),
                        };
                        match reconnected {
                            Ok(c) => {
                                client = c;
                                for p in batch {
                                    let r = Self::one(&mut client, &p).await;
                                    let _ = p.reply.send(r);
                                }
                            }
                            Err(e2) => {
                                for p in batch {
                                    let _ = p.reply.send(Err(anyhow::anyhow!("vocoder reconnect failed: {}", e2)));
                                }
                            }
                        }
                    }
                }
            }
        });
        Self { tx }
    }

    async fn one(client: &mut WorkerClient, p: &Pending) -> anyhow::Result<Vec<i16>> {
        let resp = client.call(&Request::Vocode { codes: p.codes.clone(), n_tokens: p.n_tokens }).await?;
        match &resp.data {
            ResponseData::Vocode { audio, .. } => decode_i16(audio),
            _ => anyhow::bail!("Unexpected vocoder response"),
        }
    }

    /// Queue `n_tokens` frames of codes; the receiver yields their PCM.  Never blocks (the fire-and-forget
    /// `send_request` of pipeline.rs:596-602 becomes this).
    pub fn submit(&self, codes: Vec<u8>, n_tokens: usize) -> oneshot::Receiver<anyhow::Result<Vec<i16>>> {
        let (reply, rx) = oneshot::channel();
        if let Err(mpsc::error::SendError(p)) = self.tx.send(Pending { codes, n_tokens, reply }) {
            let _ = p.reply.send(Err(anyhow::anyhow!("vocoder queue closed")));
        }
        rx
    }
}

// ---- how the call sites change -------------------------------------------------------------------------------------------
//
// struct Pipeline { ..., vocoder: VocoderQueue }          // was: vocoder: WorkerClient   (pipeline.rs:126-135)
// Pipeline::new:  vocoder: VocoderQueue::spawn(WorkerClient::connect(&cfg.vocoder.host, cfg.vocoder.port).await?, 256)
//
// synthesize_single, the streaming send (pipeline.rs:586-607):
//     let mut voc_rx = Vec::new();
//     ...
//     if all_codes.len() >= (vocode_batches_sent + 1) * vocode_chunk_size {
//         ...
//         voc_rx.push(self.vocoder.submit(encode_i64(&batch_codes), batch_n));
//         vocode_batches_sent += 1;
//     }
// the tail (pipeline.rs:644-657):
//     if sent_tokens < n_tokens { voc_rx.push(self.vocoder.submit(encode_i64(&remaining_codes), remaining_n)); }
// and the collection loop (pipeline.rs:659-673), submission order kept by awaiting the receivers in order:
//     for rx in voc_rx { all_audio_i16.extend_from_slice(&rx.await??); }
//
// synthesize_pipelined (pipeline.rs:346-377): `voc_rx.push(self.vocoder.submit(encode_i64(&flat), n_chunk_tokens))` per
// text chunk, no "keep one in flight" drain (the queue does the pacing), and the same in-order await loop at the end.
//
// ---- DESIGN NOTE, not a patch ------------------------------------------------------------------------------------------------
// What follows does not exist in the reference and is NOT written out here: `Pipeline::generate` / `Pipeline::collect_audio`
// are names for the two halves a maintainer would cut `synthesize_single` (pipeline.rs:403-692) and `synthesize_pipelined`
// (pipeline.rs:180-400) into -- everything up to and including the tail submit (pipeline.rs:644-657 / 347-353) on one side, the
// in-order response loop (pipeline.rs:659-673 / 356-377) and the RTF log lines (679-684 / 385-392) on the other.  The split
// moves ~20 local variables (timers, `all_codes`, `vocode_batches_sent`) across the cut; it needs a compiler to get right, so
// it is left as this note.  The code ABOVE this line (VocoderQueue and the call-site substitutions) is the patch.
//
// api/openai.rs:86 and mcp/server.rs:243 hold `synthesis_lock` around the whole `Pipeline::synthesize`; with the split the guard
// would cover only the talker / predictor phases:
//     let guard = state.synthesis_lock.clone().lock_owned().await;          // phases 1-3: talker / predictor are single-client
//     let (codes, voc_rx) = pipeline.generate(&params).await?;               // submits vocoder work as it goes
//     drop(guard);                                                          // the next request's generation starts here
//     let audio = pipeline.collect_audio(voc_rx).await?;                     // phase 4 overlaps it
// Without the split the queue still coalesces: concurrent `speak` CLI processes and several frontends share one vocoder worker.
