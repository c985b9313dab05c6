/**
 * GroupByCombineOperator with the library merge in front (pinot-core/.../operator/combine/GroupByCombineOperator.java:100-160): when
 * every segment operator of the query is a GpuGroupByOperator, the per-segment accumulator tables stay in HBM
 * (QUERY_FLAG_KEEP_DEVICE_TABLE), tables of one GPU are folded there (pg_result_merge), the per-GPU tables are folded over RCCL
 * (pg_result_all_reduce: every rank ends with the merged table) and ONE table is decoded.  What the unchanged combine code then sees is
 * one non-empty results block and empty ones: IndexedTable upserts, trimming, ordering and the instance response are the reference's.
 *
 * The library refuses what it cannot merge by dictId — segments with different dictionaries, hashed key spaces, trimmed tables
 * (UnsupportedOperationException, decided on every rank alike before any exchange: pg_comm.cpp): the tables that did not fold are
 * decoded one by one and merged by values, which is what GroupByCombineOperator always does.
 *
 * The seam: CombinePlanNode#getCombineOperator (core/plan/CombinePlanNode.java:140-145) constructs GroupByCombineOperator itself and
 * InstancePlanMakerImplV2#makeInstancePlan (:171-198) keeps its option handling private, so a plan maker cannot substitute the combine
 * operator; INTEGRATION.md §4.2 shows the four-line patch of getCombineOperator that constructs this class.
 */
package org.apache.pinot.gpu;

import java.util.ArrayList;
import java.util.List;
import java.util.Map;
import java.util.TreeMap;
import java.util.concurrent.CancellationException;
import java.util.concurrent.ExecutionException;
import java.util.concurrent.atomic.AtomicReferenceArray;
import java.util.concurrent.ExecutorService;
import java.util.concurrent.Future;
import org.apache.pinot.core.common.Operator;
import org.apache.pinot.core.operator.blocks.results.BaseResultsBlock;
import org.apache.pinot.core.operator.combine.GroupByCombineOperator;
import org.apache.pinot.core.query.request.context.QueryContext;

public class GpuGroupByCombineOperator extends GroupByCombineOperator {
  private static final String EXPLAIN_NAME = "GPU_COMBINE_GROUP_BY";

  // A cross-GPU merge holds ONE communicator set for its duration (GpuInstancePlanMaker#acquireCommunicators): pg_result_all_reduce keeps
  // per-communicator state (probe, scratch) and every rank must enter the collectives of one merge on the same set — two merges on one set
  // would interleave their collectives differently on different ranks, which RCCL leaves as a hang (the library refuses loudly when a
  // communicator is entered twice: pg_comm.cpp).  A pool of K sets lets K merges proceed at once (VERDICT r5 #9); the set is handed back only
  // after every rank's call has returned.

  private final List<Operator> _segmentOperators;
  private final ExecutorService _workers;

  public GpuGroupByCombineOperator(List<Operator> operators, QueryContext queryContext, ExecutorService executorService) {
    super(operators, queryContext, executorService);
    _segmentOperators = operators;
    _workers = executorService;
  }

  /** CombinePlanNode's test: the library merge applies when every segment runs on the accelerated path. */
  public static boolean applies(List<Operator> operators) {
    if (GpuInstancePlanMaker.current() == null || operators.size() < 2) {
      return false;
    }
    for (Operator operator : operators) {
      if (!(operator instanceof GpuGroupByOperator)) {
        return false;
      }
    }
    return true;
  }

  @Override
  public String toExplainString() {
    return EXPLAIN_NAME;
  }

  @Override
  protected BaseResultsBlock getNextBlock() {
    boolean folded = applies(_segmentOperators);
    if (folded) {
      foldInLibrary();
    }
    try {
      return super.getNextBlock();   // upserts the (now mostly empty) segment blocks into the IndexedTable, trims, orders
    } finally {
      // a combine that stopped early (time-out, cancellation, another segment's exception) never asked some operators for their block:
      // their adopted native results — HBM tables included — are freed here, not leaked (ADVICE r4)
      if (folded) {
        for (Operator operator : _segmentOperators) {
          ((GpuGroupByOperator) operator).releaseAdopted();
        }
      }
    }
  }

  private void foldInLibrary() {
    int n = _segmentOperators.size();
    long[] results = new long[n];
    try {
      foldInLibrary(results);
    } catch (RuntimeException e) {   // cancelled (EarlyTerminationException) or failed midway: nothing native may outlive the query
      for (int i = 0; i < n; i++) {
        if (results[i] != 0) {
          PinotGpu.resultFree(results[i]);
          results[i] = 0;
        }
      }
      throw e;
    }
  }

  private void foldInLibrary(long[] results) {
    GpuInstancePlanMaker maker = GpuInstancePlanMaker.current();
    int n = _segmentOperators.size();
    GpuGroupByOperator[] ops = new GpuGroupByOperator[n];
    // one worker task per segment, as BaseCombineOperator runs them (BaseCombineOperator.java:97-142): the segments' queries overlap on
    // their GPUs' streams.  EVERY task is waited for — through interrupts too — before anything is thrown, so no native call outlives this
    // method and every result that came back sits in results[] for the caller to free.  The cancel tokens of the running queries are
    // registered under the POOL threads (GpuGroupByOperator#execute), which the query killer does not know: when it interrupts THIS thread
    // (time-out, kill), the interrupt is handed on to them — GpuCancellation.cancel(runner) — and the tasks that have not started are cancelled.
    List<Future<Long>> executions = new ArrayList<>(n);
    AtomicReferenceArray<Thread> runners = new AtomicReferenceArray<>(n);
    for (int i = 0; i < n; i++) {
      ops[i] = (GpuGroupByOperator) _segmentOperators.get(i);
      GpuGroupByOperator op = ops[i];
      int slot = i;
      executions.add(_workers.submit(() -> {   // 0: refused at run time (hash bucket overflow) — that operator answers with its Java plan
        runners.set(slot, Thread.currentThread());
        try {
          return op.execute();
        } finally {
          runners.set(slot, null);
        }
      }));
    }
    RuntimeException failed = null;
    boolean interrupted = false;
    for (int i = 0; i < n; i++) {
      for (;;) {
        try {
          results[i] = executions.get(i).get();
          break;
        } catch (ExecutionException e) {
          if (failed == null) {
            failed = e.getCause() instanceof RuntimeException ? (RuntimeException) e.getCause() : new RuntimeException(e.getCause());
          }
          break;
        } catch (CancellationException e) {   // cancelled below before it started: nothing ran, nothing to free
          break;
        } catch (InterruptedException e) {
          if (!interrupted) {
            interrupted = true;
            for (int j = 0; j < n; j++) {
              executions.get(j).cancel(false);           // not started yet: never will; running: left to its cancel token
              Thread runner = runners.get(j);
              if (runner != null) {
                GpuCancellation.cancel(runner);          // EarlyTerminationException inside pg_query_exec: the task ends promptly
              }
            }
          }
          // keep waiting: the running tasks own native results and HBM tables until they return
        }
      }
    }
    if (interrupted) {
      Thread.currentThread().interrupt();
      if (failed == null) {
        failed = new RuntimeException(new InterruptedException("interrupted while the segments' GPU queries were running"));
      }
    }
    if (failed != null) {
      throw failed;   // (the caller frees what did come back)
    }
    // tables of one GPU fold into the first table of that GPU; a table the library refuses to fold stays a head of its own
    Map<Integer, List<Integer>> headsOfDevice = new TreeMap<>();
    for (int i = 0; i < n; i++) {
      if (results[i] == 0 || !ops[i].tableKept()) {   // refused, or answered without a device table: merges by values below
        continue;
      }
      List<Integer> heads = headsOfDevice.computeIfAbsent(ops[i].device(), d -> new ArrayList<>());
      boolean folded = false;
      for (int head : heads) {
        try {
          PinotGpu.resultMerge(results[head], results[i]);
          folded = true;
          break;
        } catch (UnsupportedOperationException differentDictionaries) {
          // pg_result_merge checks the layout signature (dictionary contents included) before it touches either table
        }
      }
      if (folded) {
        PinotGpu.resultFree(results[i]);
        results[i] = 0;
        ops[i].adoptFolded();
      } else {
        heads.add(i);
      }
    }
    // one table per GPU of the communicator: fold across the GPUs; every rank calls from a thread of its own (the collective returns
    // when all ranks joined), and every rank ends with the merged table — rank 0's is decoded, the others are dropped
    boolean onePerDevice = headsOfDevice.size() == maker.numDevices() && headsOfDevice.size() > 1;
    for (List<Integer> heads : headsOfDevice.values()) {
      onePerDevice &= heads.size() == 1;
    }
    if (onePerDevice) {
      List<Integer> ranks = new ArrayList<>();
      List<Future<?>> calls = new ArrayList<>();
      boolean reduced = true;
      RuntimeException collectiveFailed = null;
      long[] set;
      try {
        set = maker.acquireCommunicators();   // blocks while every set is inside a merge
      } catch (InterruptedException e) {
        Thread.currentThread().interrupt();
        throw new RuntimeException(e);        // nothing has entered a collective yet: the caller frees the results
      }
      try {
        for (Map.Entry<Integer, List<Integer>> e : headsOfDevice.entrySet()) {
          int head = e.getValue().get(0);
          long comm = set[maker.indexOfDevice(e.getKey())];
          ranks.add(head);
          calls.add(_workers.submit(() -> PinotGpu.resultAllReduce(results[head], comm)));
        }
        boolean interrupted = false;
        for (Future<?> call : calls) {   // EVERY rank's call is waited for — through interrupts too — before the set goes back and before anything is thrown
          for (;;) {
            try {
              call.get();
              break;
            } catch (ExecutionException e) {
              if (!(e.getCause() instanceof UnsupportedOperationException)) {   // the refusal is collective: every rank threw it
                if (collectiveFailed == null) {
                  collectiveFailed = new RuntimeException(e.getCause());
                }
              }
              reduced = false;
              break;
            } catch (InterruptedException e) {
              interrupted = true;   // the ranks are inside a collective: leaving now would strand them (and the communicators)
            }
          }
        }
        if (interrupted) {
          Thread.currentThread().interrupt();
          reduced = false;
          if (collectiveFailed == null) {
            collectiveFailed = new RuntimeException(new InterruptedException("interrupted during the cross-GPU merge"));
          }
        }
      } finally {
        maker.releaseCommunicators(set);
      }
      if (collectiveFailed != null) {
        throw collectiveFailed;
      }
      if (reduced) {
        for (int r = 1; r < ranks.size(); r++) {
          int i = ranks.get(r);
          PinotGpu.resultFree(results[i]);
          results[i] = 0;
          ops[i].adoptFolded();
        }
      }
    }
    for (int i = 0; i < n; i++) {
      if (results[i] != 0) {
        ops[i].adopt(results[i]);   // decoded (and freed) by the operator's getNextBlock inside the unchanged combine loop
      }
    }
  }
}
