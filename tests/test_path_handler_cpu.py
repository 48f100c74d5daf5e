"""sortham::PathHandler and sortham::TrajectoryVisualizer for plain types
against the value assertions of the reference's own tests: test/path_handler_test.cpp,
test/trajectory_visualizer_tests.cpp and the inversion helpers of test/utils_test.cpp:384-445.
No GPU involved (the host library loads without one)."""
import numpy as np
import pytest

from mpcholonavigation_amd.path_handler import (PathHandler, TrajectoryVisualizer, find_first_path_inversion,
                                                remove_poses_after_first_inversion)


def _line(n, x0=0.0):
    p = np.zeros((n, 3))
    p[:, 0] = x0 + np.arange(n)
    return p


def test_get_and_prune_path():
    """path_handler_test.cpp:85-103 — 11 poses, pruned up to begin + 5: 6 left."""
    h = PathHandler()
    h.set_path(np.zeros((11, 3)))
    assert len(h.get_path()) == 11
    h.prune(5)
    assert len(h.get_path()) == 6


def test_bounds():
    """path_handler_test.cpp:105-156 — default costmap (100 x 100 cells at 0.05 m): max costmap
    distance 2.5; 100 poses x = i, robot at x = 25: the closest pose is index 25 and pruning up to
    it leaves 75."""
    h = PathHandler(costmap_size=(100, 100), resolution=0.05, max_robot_pose_search_dist=99999.9)
    assert h.max_costmap_dist() == 2.5
    h.set_path(_line(100))
    plan, closest = h.plan_in_bounds((25.0, 0.0, 0.0))
    assert closest == 25
    h.prune(closest, up_to_inversion=True)
    assert len(h.get_path(up_to_inversion=True)) == 75


def test_transforms_and_transform_path():
    """path_handler_test.cpp:158-213 — transformPath throws on an empty plan; with a plan it
    returns what getGlobalPlanConsideringBoundsInCostmapFrame returns."""
    h = PathHandler(costmap_size=(100, 100), resolution=0.05, max_robot_pose_search_dist=99999.9)
    with pytest.raises(RuntimeError, match="Received plan with zero length"):
        h.transform_path((2.5, 0.0, 0.0))
    h.set_path(_line(100))
    path_out, closest = h.plan_in_bounds((2.5, 0.0, 0.0))
    final = h.transform_path((2.5, 0.0, 0.0))
    assert len(final) == len(path_out) >= 1
    # the plan is cut at the costmap's edge (5 m x 5 m from the origin) and at prune_distance
    assert np.all(final[:, 0] < 5.0)
    # a rigid transform between the plan's frame and the costmap's is applied pose by pose
    h2 = PathHandler(costmap_size=(200, 200), resolution=0.05, prune_distance=1.5)
    pts = np.zeros((40, 3))
    pts[:, 0] = 0.1 * np.arange(40)
    h2.set_path(pts)
    tf = (1.0, 2.0, np.pi / 2)
    out = h2.transform_path((0.0, 0.0, 0.0), tf)
    assert np.allclose(out[:, 0], 1.0, atol=1e-12) and np.allclose(out[:, 1], 2.0 + 0.1 * np.arange(len(out)))
    assert np.allclose(out[:, 2], np.pi / 2)
    # prune_distance 1.5 at 0.1 m spacing: the first pose beyond 1.5 m of path is index 16
    assert len(out) == 16
    g = h2.transformed_goal(tf)
    assert np.allclose(g, (1.0, 2.0 + 3.9, np.pi / 2))
    # a plan that leaves the costmap at its very first pose: the reference throws
    h3 = PathHandler(costmap_size=(10, 10), resolution=0.05)
    h3.set_path(_line(5, x0=100.0))
    with pytest.raises(RuntimeError, match="Resulting plan has 0 poses in it."):
        h3.transform_path((100.0, 0.0, 0.0))


def _yaw(z, w):
    return 2.0 * np.arctan2(z, w)     # tf2::getYaw of a rotation about z


def test_inversion_tolerance_checks():
    """path_handler_test.cpp:215-255."""
    h = PathHandler()
    path = _line(10)
    h.set_path(path)
    assert not h.within_inversion_tolerances((0.0, 0.0, 0.0))            # not near the last pose
    assert h.within_inversion_tolerances((9.0, 0.0, 0.0))                # exactly on top of it
    assert not h.within_inversion_tolerances((9.0, 9.0, 0.0))            # laterally off
    assert not h.within_inversion_tolerances((9.0, 0.0, _yaw(0.8509035, 0.525322)))    # off angled
    assert h.within_inversion_tolerances((9.0, 0.0, _yaw(0.0871558, 0.9961947)))       # within tolerances
    assert h.within_inversion_tolerances((9.10, 0.0, _yaw(0.0871558, 0.9961947)))      # offset + angled, both within


def test_find_path_inversion():
    """utils_test.cpp:384-411."""
    assert find_first_path_inversion(_line(10)) == 10
    assert find_first_path_inversion(_line(10)[7:]) == 3             # too short to process
    cusp = np.zeros((20, 3))
    cusp[:10, 0] = np.arange(10)
    cusp[10:, 0] = 10 - np.arange(10)
    assert find_first_path_inversion(cusp) == 11


def test_remove_poses_after_path_inversion():
    """utils_test.cpp:413-445."""
    r, p = remove_poses_after_first_inversion(_line(10))
    assert r == 0 and len(p) == 10
    r, p = remove_poses_after_first_inversion(np.zeros((0, 3)))
    assert r == 0 and len(p) == 0
    cusp = np.zeros((20, 3))
    cusp[:10, 0] = np.arange(10)
    cusp[10:, 0] = 10 - np.arange(10)
    r, p = remove_poses_after_first_inversion(cusp)
    assert r == 11 and len(p) == 11 and p[-1, 0] == 10


def test_enforced_inversion_keeps_the_first_leg_until_reached():
    """setPath with enforce_path_inversion (path_handler.cpp:173-180) crops the working plan at the
    cusp; transformPath releases the rest once the robot is within the inversion tolerances
    (:133-139)."""
    cusp = np.zeros((20, 3))
    cusp[:10, 0] = np.arange(10) * 0.1
    cusp[10:, 0] = (10 - np.arange(10)) * 0.1
    h = PathHandler(costmap_size=(200, 200), resolution=0.05, origin=(-5.0, -5.0), enforce_path_inversion=1)
    h.set_path(cusp)
    assert len(h.get_path()) == 20 and len(h.get_path(up_to_inversion=True)) == 11
    out = h.transform_path((0.0, 0.0, 0.0))
    assert len(out) == 11 and out[-1, 0] == pytest.approx(1.0)
    out = h.transform_path((1.0, 0.0, 0.0))          # at the cusp: the second leg is released
    assert len(h.get_path()) == 9                    # pruned up to the inversion
    assert len(h.get_path(up_to_inversion=True)) == 9


def test_visualizer_optimal_trajectory():
    """trajectory_visualizer_tests.cpp:68-129."""
    vis = TrajectoryVisualizer("fkmap")
    vis.add_trajectory(np.zeros((0, 2), np.float32))       # empty: nothing to publish
    assert vis.visualize() == []
    vis.add_trajectory(np.ones((20, 2), np.float32))
    ms = vis.visualize()
    assert len(ms) == 20 and ms[0]["frame_id"] == "fkmap"
    assert [ms[i]["id"] for i in (0, 1, 10)] == [0, 1, 10]
    assert ms[0]["position"] == (1.0, 1.0, 0.06)
    assert ms[0]["scale"] == (0.03, 0.03, 0.07)
    assert ms[19]["scale"] == (0.07, 0.07, 0.09)
    for a, b in zip(ms[:-1], ms[1:]):
        assert a["color"][1] < b["color"][1] and a["color"][2] < b["color"][2]
        assert a["color"][0] == b["color"][0] and a["color"][3] == b["color"][3]
    assert vis.visualize() == []                            # visualize() resets


def test_visualizer_candidate_trajectories():
    """trajectory_visualizer_tests.cpp:131-155 — 200 x 12 with trajectory_step 5, time_step 3: 40 * 4."""
    vis = TrajectoryVisualizer("fkmap")
    ones = np.ones((200, 12), np.float32)
    vis.add_candidates(ones, ones)
    assert len(vis.visualize()) == 160
